#!/bin/bash
# A/B timing of library variants built into ray_tracing_amd/csrc/variants (book-1 scene, 4K, 500 spp).
set -e
for v in "" $(ls ray_tracing_amd/csrc/variants/*.so 2>/dev/null); do
  for blk in 256 512; do
    echo "== ${v:-default} block $blk"
    RT_HIP_LIB=$v RT_CHUNK=2 timeout -k 10 120 python -u tools/quick_bench.py 3840 500 50 $blk 2>&1 | tail -1
  done
done
