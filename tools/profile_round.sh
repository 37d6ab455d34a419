#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...
# kernel-trace stats of the bench command (pipelined and one-step-at-a-time), separate --pmc passes
# (never combined with trace domains), and the FETCH_SIZE calibration microbenchmark.
set -e
TAG=${1:-prof}; DEPTH=${2:-10000}; STEPS=${3:-20}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_pipelined -o b --output-format csv -- python3 $ROOT/bench.py --depth $DEPTH --steps $STEPS --warmup 3 --cpu-passes -1 --no-extra --no-e2e > $OUT/kt_pipelined.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_serial -o b --output-format csv -- python3 $ROOT/bench.py --depth $DEPTH --steps $STEPS --warmup 3 --cpu-passes -1 --no-pipeline --no-extra --no-e2e > $OUT/kt_serial.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 10 200 rocprofv3 --pmc $C -d $OUT/pmc_$N -o p --output-format csv -- python3 $ROOT/bench.py --depth $DEPTH --steps 4 --warmup 1 --cpu-passes -1 --no-pipeline --no-extra --no-e2e > $OUT/pmc_$N.log 2>&1
  echo "pmc $N done"
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $ROOT/tools/micro/fetch_calib.hip > $OUT/calib_build.log 2>&1
timeout -k 10 100 rocprofv3 --pmc FETCH_SIZE -d $OUT/calib_fetch -o p --output-format csv -- /tmp/fetch_calib > $OUT/calib_fetch.log 2>&1
timeout -k 10 100 rocprofv3 --pmc WRITE_SIZE -d $OUT/calib_write -o p --output-format csv -- /tmp/fetch_calib > $OUT/calib_write.log 2>&1
timeout -k 10 100 rocprofv3 --kernel-trace --stats -d $OUT/calib_kt -o p --output-format csv -- /tmp/fetch_calib > $OUT/calib_kt.log 2>&1
echo "profile_round done"
