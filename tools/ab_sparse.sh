#!/bin/bash
# A/B of code objects on the two scenes whose bounce scan has block boxes: tools/ab_sparse.sh a.hsaco b.hsaco ...
set -e
export RT_CHUNK=1
for h in "$@"; do
  echo "== $h"
  RT_HSACO=$h timeout -k 10 200 python -u tools/quick_bench.py 3840 500 50 0 book1 2>&1 | tail -2 | sed 's/.*kernel/kernel/; s/crc.*rt_/rt_/'
  RT_HSACO=$h timeout -k 10 300 python -u tools/quick_bench.py 1920 64 16 0 grid 2>&1 | tail -2 | sed 's/.*kernel/kernel/; s/crc.*rt_/rt_/'
done
