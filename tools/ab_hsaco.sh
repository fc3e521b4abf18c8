#!/bin/bash
# A/B timing of code objects on ONE box (devices differ by several per cent, so both arms must run in the same
# call): tools/ab_hsaco.sh a.hsaco b.hsaco ...   -- each is timed twice, interleaved, on the four configurations.
set -e
export RT_CHUNK=${RT_CHUNK:-1}
for round in 1 2; do
  for h in "$@"; do
    echo "== round $round $h"
    RT_HSACO=$h timeout -k 10 200 python -u tools/quick_bench.py 3840 500 50 0 book1 2>&1 | tail -1 | sed 's/.*kernel/kernel/'
    RT_HSACO=$h timeout -k 10 200 python -u tools/quick_bench.py 7680 256 100 0 hall 2>&1 | tail -1 | sed 's/.*kernel/kernel/'
    RT_HSACO=$h timeout -k 10 300 python -u tools/quick_bench.py 1920 64 16 0 grid 2>&1 | tail -1 | sed 's/.*kernel/kernel/'
  done
done
