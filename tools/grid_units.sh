#!/bin/bash
# The 16,385-sphere frame and a rank's share of it at several pixel-unit shapes, one box: tools/grid_units.sh [shapes...]
set -e
export RT_SCENE=grid RT_CHUNK=1
echo default; RT_DEBUG_LAUNCH=1 timeout -k 10 120 python3 tools/scale_probe.py 64 1,2,4,8 2>&1 | grep -E "world|rt launch" | uniq
for u in ${@:-8x1 4x2 4x1 2x2}; do echo unit $u; RT_UNIT=$u timeout -k 10 120 python3 tools/scale_probe.py 64 1,4,8 2>&1 | grep -E "^world"; done
