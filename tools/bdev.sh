#!/bin/bash
# development build of libamplihip.so with extra -D flags into amplipy_amd/build/libamplihip_dev.so (use: AMPLIHIP_LIB=amplipy_amd/build/libamplihip_dev.so)
set -e
cd "$(dirname "$0")/.."
mkdir -p amplipy_amd/build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c -o amplipy_amd/build/amplihip_dev.o amplipy_amd/csrc/amplihip.hip > /tmp/build_dev.log 2>&1 || { grep -B2 -A8 "error" /tmp/build_dev.log | head -60; echo BUILD FAILED; exit 1; }
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o amplipy_amd/build/libamplihip_dev.so amplipy_amd/build/amplihip_dev.o amplipy_amd/build/amp_ins.o -ldl
echo DEV BUILD OK
