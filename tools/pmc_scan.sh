#!/bin/bash
# counter passes over the scan kernels of the bench workload (development aid; run through gpurun from the repo root):
#   tools/pmc_scan.sh <tag> "<counters>" ["<counters>" ...]      -> gpurun_out/<tag>/pmc_x<i>/
# Each pass is a separate rocprofv3 --pmc run (never combined with a trace domain).  AMPLIHIP_LIB / SCAN_ARGS are passed on.
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
I=0
for C in "$@"; do
  I=$((I+1))
  timeout -k 10 150 rocprofv3 --pmc $C -d $OUT/pmc_x$I -o p --output-format csv -- python3 $ROOT/tools/run_scan.py --iters 4 $SCAN_ARGS > $OUT/pmc_x$I.log 2>&1 || echo "pmc $C failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT 2>/dev/null | tee $OUT/summary.txt
echo done
