#!/bin/bash
# Spread of the HBM write counter over launches and processes, on one box: tools/pmc_write_spread.sh <workload>
# (per dispatch of the render kernel, KiB; three processes of three launches each)
set -e
WL=${1:-c3}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_spread
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w$i" -- python3 "$REPO/bench.py" --workload "$WL" --steps ${STEPS:-2} --warmup ${WARMUP:-1} --no-cpu-baseline --e2e-steps 0 --brute-steps 0 > "$OUT/w$i.log" 2>&1
  python3 - "$OUT/w$i" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_render" in r["Kernel_Name"]:
            print(r["Kernel_Name"], "dispatch", r["Dispatch_Id"], r["Counter_Name"], float(r["Counter_Value"]), "KiB")
PY
done
