#!/bin/bash
# Per-phase instruction census of k_tile (development aid): AMPLIHIP_PHASES masks phases off
# (1 = P1/P3 only, +2 = P2 quality windows, +4 = P4 base counting); results are wrong on purpose.
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-phases}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PH in 0x1 0x3 0x5 0xFF; do
  export AMPLIHIP_PHASES=$PH
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d $OUT/pmc_$PH -o p --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-passes 0 --no-pipeline > $OUT/pmc_$PH.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$PH -o p --output-format csv -- python3 $ROOT/bench.py --steps 5 --warmup 1 --cpu-passes 0 --no-pipeline > $OUT/kt_$PH.log 2>&1
  echo "phases $PH done"
done
