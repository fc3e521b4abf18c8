#!/bin/bash
# Counters of the per-wavefront and the workgroup-pool schedulers side by side on one box:
#   bash tools/pmc_pool.sh <tag> "<quick_bench args>" <schedulers...>     (wave = per-wavefront units, pool, auto)
set -e
TAG=$1; ARGS=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_pool_$TAG
rm -rf "$OUT" "$OUT.txt"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp RT_CHUNK=${RT_CHUNK:-1}
for n in "$@"; do
  export RT_SCHED=$n
  timeout -s KILL 200 rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE} \
      --output-format csv -d "$OUT/$n" -- python3 "$REPO/tools/quick_bench.py" $ARGS > "$OUT/$n.log" 2>&1
  echo "== $n" >> "$OUT.txt"
  tail -1 "$OUT/$n.log" | sed 's/.*kernel/kernel/' >> "$OUT.txt"
  python3 "$REPO/tools/pmc_summary.py" "$OUT/$n" rt_render >> "$OUT.txt"
done
cat "$OUT.txt"
