#!/bin/bash
# Marginal cost of parts of the render loop: code objects that do one part twice (same image), timed on
# the 4K / 500 spp book frame.  make -C ray_tracing_amd/csrc experiments   first.
set -e
for v in "" ray_tracing_amd/csrc/rt_kernels_x_*.hsaco; do
  echo "== ${v:-baseline}"
  RT_HSACO=$v RT_CHUNK=${RT_CHUNK:-1} timeout -k 10 200 python -u tools/quick_bench.py ${RT_ARGS:-3840 500 50 0 book1} 2>&1 | tail -2
done
