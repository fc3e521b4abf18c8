#!/bin/bash
# A/B of code objects on the four benchmark frames, one box: tools/ab_all.sh a.hsaco b.hsaco ...
set -e
export RT_CHUNK=1
for h in "$@"; do
  echo "== $h"
  for a in "3840 500 50 0 book1" "7680 256 100 0 hall" "1920 64 16 0 grid" "1920 100 50 0 cpu"; do
    RT_HSACO=$h timeout -k 10 300 python -u tools/quick_bench.py $a 2>&1 | tail -2 | sed 's/.*kernel/kernel/; s/crc.*rt_/rt_/'
  done
done
