#!/bin/bash
# config 5 through development builds of the library (tools/bdev_tag.sh): tools/abl_c5.sh <variant> <tag> ...
v=$1; shift
for t in "$@"; do
  echo "== $t"
  AMPLIPY_DEV=1 AMPLIHIP_LIB=amplipy_amd/build/dev_$t.so AMP_VARIANT=$v timeout -k 10 200 python tools/time_config5.py 200 --no-check 2>&1 | grep "launch 2"
done
