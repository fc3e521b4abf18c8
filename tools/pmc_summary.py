"""Sums rocprofv3 --pmc CSV output per kernel: python tools/pmc_summary.py <dir-or-csv> [kernel-substring]"""
import collections, csv, glob, os, sys
path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "rt_"
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
if len(files) > 1:                                   # a directory that has collected several runs: the newest only
    files = [max(files, key=os.path.getmtime)]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
meta = {}
for f in files:
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
for k, v in agg.items():
    n = len(disp[k])
    print(f"kernel {k}: {n} dispatches; VGPR {meta[k][0]} SGPR {meta[k][1]} LDS {meta[k][2]} WG {meta[k][3]} grid {meta[k][4]}")
    for c, x in sorted(v.items()):
        print(f"  {c:32s} {x / n:18.1f} per dispatch")
