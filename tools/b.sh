#!/bin/bash
# build libamplihip.so (+ resource usage of the named kernel); fails loudly
set -e
cd "$(dirname "$0")/.."
K=${1:-_ZN3amp6k_fastILi4}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage \
  -o amplipy_amd/libamplihip.so amplipy_amd/csrc/amplihip.hip -ldl > /tmp/build.log 2>&1 || { grep -B2 -A8 "error" /tmp/build.log | head -60; echo BUILD FAILED; exit 1; }
grep -A11 "Function Name: $K" /tmp/build.log | grep "VGPRs:\|Scratch\|Spill\|Occupancy\|LDS" | sed 's/.*remark: *//'
echo BUILD OK
