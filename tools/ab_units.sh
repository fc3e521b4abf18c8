#!/bin/bash
# Unit-shape timing: full 4K frame and one eighth of it (the share of one of 8 GPUs).
set -e
for u in 8x1 4x1 16x1 2x2 4x2; do
  echo "== unit $u"
  RT_UNIT=$u RT_CHUNK=2 timeout -k 10 200 python -u tools/scale_probe.py 500 1,8 2>&1 | tail -2
done
