#!/bin/bash
# Profiling recipe of the render kernel, to be run on the GPU box from the repository root:
#   bash tools/profile_recipe.sh r02 c3      -> gpurun_out/prof_r02_c3/{trace,pmc_a..e}
# then, back in the container:  python tools/make_profile_summary.py gpurun_out/prof_r02_c3 r02_c3
# The second argument is bench.py's --workload (c2, c3, c4, c5).  One rocprofv3 run per counter group (the
# counter blocks cannot be split over passes), each under its own time limit; the kernel-trace run is the bench
# command whose JSON line the summary quotes.  The program itself follows `--` (no env, no shell in between).
set -e
TAG=${1:-r02}
WL=${2:-c3}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
sha256sum "$REPO/ray_tracing_amd/csrc/rt_kernels.hsaco" | cut -d' ' -f1 > "$OUT/code_object.sha256"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --workload "$WL" --steps 3 --warmup 1 --no-cpu-baseline --e2e-steps 0 --brute-steps 0 > "$OUT/trace.log" 2>&1
pass() {  # name, counters...
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$REPO/bench.py" --workload "$WL" --steps 1 --warmup 0 --no-cpu-baseline --e2e-steps 0 --brute-steps 0 > "$OUT/pmc_$name.log" 2>&1
}
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass b SQ_INSTS_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass c FETCH_SIZE
pass d WRITE_SIZE
pass e SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INSTS_FLAT
grep '^{' "$OUT/trace.log" | tail -1
