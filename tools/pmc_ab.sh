#!/bin/bash
# Instruction and cycle counters of code objects side by side on ONE box:
#   bash tools/pmc_ab.sh <tag> <quick_bench args in quotes> a.hsaco b.hsaco ...
# One rocprofv3 --pmc pass per code object (8 SQ counters + GRBM_GUI_ACTIVE), results summed per kernel by
# tools/pmc_summary.py into gpurun_out/pmc_ab_<tag>.txt.  The program itself follows `--`.
set -e
TAG=$1; ARGS=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_ab_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp RT_CHUNK=${RT_CHUNK:-1}
for h in "$@"; do
  name=$(basename "$h" .hsaco)
  RT_HSACO=$REPO/$h timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
      --output-format csv -d "$OUT/$name" -- python3 "$REPO/tools/quick_bench.py" $ARGS > "$OUT/$name.log" 2>&1
  echo "== $h" >> "$OUT.txt"
  tail -1 "$OUT/$name.log" | sed 's/.*kernel/kernel/' >> "$OUT.txt"
  python3 "$REPO/tools/pmc_summary.py" "$OUT/$name" rt_render >> "$OUT.txt"
done
cat "$OUT.txt"
