#!/bin/bash
# Effective shader clock of a micro-benchmark's kernels: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / kernel time.
#   bash tools/microbench/clock_probe.sh vgpr_bank
set -e
NAME=${1:-scan_rate}
REPO=$(pwd)
OUT=$REPO/gpurun_out/clock_probe
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/$NAME" -- "$REPO/tools/microbench/$NAME" > "$OUT/$NAME.log" 2>&1
python3 - "$OUT" "$NAME" <<'PY'
import csv, glob, sys, collections
name = sys.argv[2]
f = glob.glob(f"{sys.argv[1]}/{name}/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Dispatch_Id"]
    d = rows.setdefault(k, {"k": r["Kernel_Name"][:28], "grid": r["Grid_Size"], "c": 0.0, "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d["c"] += float(r["Counter_Value"])
for k, d in rows.items():
    if d["t"] > 1e6:
        print(f"{name} {d['k']:28s} grid {d['grid']:>8s} {d['t'] / 1e6:8.2f} ms  clock {d['c'] / 8 / d['t']:.3f} GHz  {d['c'] / 8 / 1e6:.1f} Mcycles")
PY
