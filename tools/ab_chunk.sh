#!/bin/bash
# sample_chunk sweep on one box (book-1 frame, 4K, 500 spp)
set -e
for round in 1 2; do for c in 1 2 4 8; do echo "== chunk $c"; RT_CHUNK=$c timeout -k 10 200 python -u tools/quick_bench.py 3840 500 50 0 book1 2>&1 | tail -1 | sed 's/.*kernel/kernel/'; done; done
