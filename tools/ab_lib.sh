#!/bin/bash
# A/B timing of whole library builds on ONE box: tools/ab_lib.sh a.so b.so ...  (each twice, interleaved).
set -e
export RT_CHUNK=${RT_CHUNK:-1}
for round in 1 2; do
  for l in "$@"; do
    echo "== round $round $l"
    RT_HIP_LIB=$l timeout -k 10 200 python -u tools/quick_bench.py 3840 500 50 0 book1 2>&1 | tail -1 | sed 's/.*kernel/kernel/'
    RT_HIP_LIB=$l timeout -k 10 200 python -u tools/quick_bench.py 7680 256 100 0 hall 2>&1 | tail -1 | sed 's/.*kernel/kernel/'
    RT_HIP_LIB=$l timeout -k 10 300 python -u tools/quick_bench.py 1920 64 16 0 grid 2>&1 | tail -1 | sed 's/.*kernel/kernel/'
  done
done
