#!/bin/bash
# HBM bytes of code objects on one box: tools/pmc_hbm.sh "<quick_bench args>" a.hsaco b.hsaco ...
# (FETCH_SIZE and WRITE_SIZE in passes of their own; KiB per dispatch of the render kernel, FETCH_SIZE before the x2 gfx950 correction)
set -e
ARGS=$1; shift
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp RT_CHUNK=1
for h in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=$REPO/gpurun_out/pmc_hbm/$(basename $h .hsaco)_$ctr
    rm -rf "$out"; mkdir -p "$out"
    RT_HSACO=$REPO/$h timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d "$out" -- python3 "$REPO/tools/quick_bench.py" $ARGS > "$out.log" 2>&1
    echo "$h $ctr $(python3 "$REPO/tools/pmc_summary.py" "$out" rt_render | grep $ctr)"
  done
done
