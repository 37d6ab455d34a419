#!/bin/bash
# rocprofv3 evidence for the scan kernels on the bench workload (run through gpurun from the repo root):
#   tools/prof_scan.sh <tag> [run_scan.py args]   -> gpurun_out/<tag>/{kt,pmc_*}
set -e
TAG=${1:-prof}; shift || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- python3 $ROOT/tools/run_scan.py "$@" > $OUT/kt.log 2>&1
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE WRITE_SIZE; do
  N=$(echo $C | tr ' ' '_' | cut -c1-20)
  rocprofv3 --pmc $C -d $OUT/pmc_$N -o p --output-format csv -- python3 $ROOT/tools/run_scan.py "$@" > $OUT/pmc_$N.log 2>&1 || echo "pmc $N failed"
done
echo "prof_scan done"
