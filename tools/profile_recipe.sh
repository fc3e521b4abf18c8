#!/bin/bash
# Profiling recipe of the render kernel, to be run on the GPU box from the repository root:
#   bash tools/profile_recipe.sh r01      -> gpurun_out/prof_r01/{trace,pmc_a..d}
# then, back in the container:  python tools/make_profile_summary.py gpurun_out/prof_r01 r01
# One rocprofv3 run per counter group (the counter blocks cannot be split over passes), each under
# its own time limit; the kernel-trace run is the bench command whose JSON line the summary quotes.
set -e
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 3 --warmup 1 > "$OUT/trace.log" 2>&1
pass() {  # name, counters...
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1
}
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass b SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass c FETCH_SIZE
pass d WRITE_SIZE
grep '^{' "$OUT/trace.log" | tail -1
