#!/bin/bash
# rocprofv3 evidence for BASELINE config 5 (8.0 M mixed reads, one launch; run through gpurun from the repo root):
#   tools/profile_config5.sh <tag> [replication]    -> gpurun_out/<tag>/  (kernel trace + separate --pmc passes)
TAG=${1:-c5prof}; REP=${2:-200}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_serial -o b --output-format csv -- python3 $ROOT/tools/time_config5.py $REP --no-check > $OUT/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 10 400 rocprofv3 --pmc $C -d $OUT/pmc_$N -o p --output-format csv -- python3 $ROOT/tools/time_config5.py $REP --no-check > $OUT/pmc_$N.log 2>&1
  echo "pmc $N done"
done
echo "profile_config5 done"
