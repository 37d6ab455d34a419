#!/bin/bash
# In-kernel phase stamps of k_fast (development aid): builds a -DAMP_DEV copy (ABL=n: ablation build, see amp_fast.hpp) of the library in a scratch copy of
# the repo (the shipped .so has no stamps) and runs tools/run_scan.py there.   usage: tools/stamps.sh [tool.py [args]]
set -e
ROOT=$(pwd); D=/tmp/amp_dev_repo
rm -rf $D; mkdir -p $D; cp -r $ROOT/amplipy_amd $ROOT/include $ROOT/tools $ROOT/oracle $D/
cd $D
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -DAMP_DEV ${ABL:+-DAMP_ABL=$ABL} ${ABLS:+-DAMP_ABL_STAMPS=1} -Wno-unused-function -o amplipy_amd/libamplihip.so amplipy_amd/csrc/amplihip.hip -ldl 2>/dev/null
T=${1:-run_scan.py}; shift || true
AMP_STAMPS=${STAMPS-1} python3 tools/$T "$@"
