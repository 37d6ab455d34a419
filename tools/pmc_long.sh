#!/bin/bash
# counter passes for the long-read workload (development aid): tools/pmc_long.sh <tag> "<counters>" ["<counters>" ...]
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
I=0
for C in "$@"; do
  I=$((I+1))
  rocprofv3 --pmc $C -d $OUT/pmc_x$I -o p --output-format csv -- python3 $ROOT/tools/time_longreads.py 10 > $OUT/pmc_x$I.log 2>&1 || echo "pmc $C failed"
done
echo done
