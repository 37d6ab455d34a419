#!/bin/bash
# development build with extra -D flags into amplipy_amd/build/dev_<tag>.so: tools/bdev_tag.sh <tag> -DX=1 ...
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p amplipy_amd/build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c -o amplipy_amd/build/dev_$tag.o amplipy_amd/csrc/amplihip.hip > /tmp/build_dev_$tag.log 2>&1 || { grep -B2 -A8 "error" /tmp/build_dev_$tag.log | head -60; echo BUILD FAILED; exit 1; }
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o amplipy_amd/build/dev_$tag.so amplipy_amd/build/dev_$tag.o amplipy_amd/build/amp_ins.o -ldl
rm -f amplipy_amd/build/dev_$tag.o
echo DEV BUILD $tag OK
