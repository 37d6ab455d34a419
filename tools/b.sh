#!/bin/bash
# build libamplihip.so (+ resource usage of the named kernel); fails loudly.  amp_ins.hip (the insertion-event sort) is
# rebuilt only when it or its header changed.
set -e
cd "$(dirname "$0")/.."
K=${1:-_ZN3amp6k_fastILi4}
mkdir -p amplipy_amd/build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
if [ ! -f amplipy_amd/build/amp_ins.o ] || [ amplipy_amd/csrc/amp_ins.hip -nt amplipy_amd/build/amp_ins.o ] || [ amplipy_amd/csrc/amp_ins.hpp -nt amplipy_amd/build/amp_ins.o ] || [ include/amplihip.h -nt amplipy_amd/build/amp_ins.o ]; then
  /opt/rocm/bin/hipcc $F -c -o amplipy_amd/build/amp_ins.o amplipy_amd/csrc/amp_ins.hip > /tmp/build_ins.log 2>&1 || { grep -B2 -A8 "error" /tmp/build_ins.log | head -60; echo BUILD FAILED; exit 1; }
fi
/opt/rocm/bin/hipcc $F -Rpass-analysis=kernel-resource-usage -c -o amplipy_amd/build/amplihip.o amplipy_amd/csrc/amplihip.hip > /tmp/build.log 2>&1 || { grep -B2 -A8 "error" /tmp/build.log | head -60; echo BUILD FAILED; exit 1; }
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o amplipy_amd/libamplihip.so amplipy_amd/build/amplihip.o amplipy_amd/build/amp_ins.o -ldl
grep -A11 "Function Name: $K" /tmp/build.log | grep "VGPRs:\|Scratch\|Spill\|Occupancy\|LDS" | sed 's/.*remark: *//'
echo BUILD OK
