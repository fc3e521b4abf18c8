#!/bin/bash
# sample_chunk = 1 on one box: item kernel (RT_NO_QUEUE=1) against the camera-ray-batch kernel (block 256 and 320),
# twice, interleaved.
set -e
export RT_CHUNK=1
for round in 1 2; do
  for cfg in "1 0" "0 0" "0 320"; do
    set -- $cfg
    echo "== round $round RT_NO_QUEUE=$1 block=$2"
    RT_NO_QUEUE=$1 timeout -k 10 200 python -u tools/quick_bench.py 3840 500 50 $2 book1 2>&1 | tail -1 | sed 's/.*kernel/kernel/'
    RT_NO_QUEUE=$1 timeout -k 10 200 python -u tools/quick_bench.py 7680 256 100 $2 hall 2>&1 | tail -1 | sed 's/.*kernel/kernel/'
    RT_NO_QUEUE=$1 timeout -k 10 200 python -u tools/quick_bench.py 1920 100 50 $2 cpu 2>&1 | tail -1 | sed 's/.*kernel/kernel/'
  done
done
