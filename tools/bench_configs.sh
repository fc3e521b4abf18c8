#!/bin/bash
# Kernel time of the five benchmark configurations on one GPU (tools/quick_bench.py, sample_chunk 1).
set -e
export RT_CHUNK=${RT_CHUNK:-1}
timeout -k 10 200 python -u tools/quick_bench.py 3840 500 50 0 book1 2>&1 | tail -1
timeout -k 10 200 python -u tools/quick_bench.py 1920 100 50 0 cpu 2>&1 | tail -1
timeout -k 10 300 python -u tools/quick_bench.py 1920 64 16 0 grid 2>&1 | tail -1
timeout -k 10 300 python -u tools/quick_bench.py 7680 256 100 0 hall 2>&1 | tail -1
