#!/bin/bash
# the bench batch through development builds of the library (tools/bdev_tag.sh): tools/abl_scan.sh "<run_scan args>" <tag> ...
args=$1; shift
for t in "$@"; do
  echo "== $t"
  AMPLIPY_DEV=1 AMPLIHIP_LIB=amplipy_amd/build/dev_$t.so timeout -k 10 200 python tools/run_scan.py $args 2>&1 | tail -3
done
