#!/bin/bash
# fast kernels on batches of uniform read length (development aid; GPU): tools/v7_sweep.sh "<lengths>" "<variants>"; AMPLIHIP_LIB / AMPLIPY_DEV are passed on
for L in ${1:-200 250 300}; do
  for V in ${2:-0 7}; do
    echo "== read-len $L variant $V ${AMPLIHIP_LIB}"; timeout -k 10 120 python tools/run_scan.py --read-len $L --variant $V --iters 5 --check 2>&1 | tail -3
  done
done
