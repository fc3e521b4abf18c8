#!/bin/bash
# HBM bytes of the launches of one rank's share of a frame: tools/pmc_scale.sh <spp> <world> [scene]
# (tools/scale_probe.py under rocprofv3, FETCH_SIZE and WRITE_SIZE in passes of their own; KiB per dispatch,
# FETCH_SIZE before the x2 gfx950 correction)
set -e
SPP=$1; WORLD=$2; export RT_SCENE=${3:-book1} RT_CHUNK=1
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$REPO/gpurun_out/pmc_scale/$ctr
  rm -rf "$out"; mkdir -p "$out"
  timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d "$out" -- python3 "$REPO/tools/scale_probe.py" $SPP $WORLD > "$out.log" 2>&1
  python3 "$REPO/tools/pmc_summary.py" "$out" rt_render
done
