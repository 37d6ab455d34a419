#!/bin/bash
# variant 7 against the default choice on uniform read lengths and on config 5 with another window (development aid; GPU)
for L in 200 250 300; do
  for V in 0 7; do
    echo "== read-len $L variant $V"; timeout -k 10 120 python tools/run_scan.py --read-len $L --variant $V --iters 5 --check 2>&1 | tail -3
  done
done
