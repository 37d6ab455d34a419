#!/bin/bash
# instruction counts of ablation builds of the fast kernel (development aid): tools/pmc_abl.sh <tag> <lib> [<lib> ...]
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LIBF in "$@"; do
  N=$(basename $LIBF .so)
  AMPLIHIP_LIB=$ROOT/$LIBF timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_WAIT_INST_ANY -d $OUT/$N -o p --output-format csv -- python3 $ROOT/tools/run_scan.py --iters 3 --variant 5 > $OUT/$N.log 2>&1 || echo "$N failed"
  echo "== $N: $(grep variant $OUT/$N.log)"
  python3 $ROOT/tools/pmc_summary.py $OUT/$N 2>/dev/null | grep -A9 "k_fast" | head -10
done
