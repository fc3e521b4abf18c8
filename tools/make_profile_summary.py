"""Condenses a gpurun_out/prof_rNN directory (rocprofv3 --kernel-trace --stats run + separate --pmc passes of
bench.py) into the files kept under profiles/:  python tools/make_profile_summary.py gpurun_out/prof_r01 r01"""
import collections, csv, glob, json, os, re, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

# 1. kernel-trace --stats summary (verbatim rows, long torch kernel names shortened)
stats = max(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
with open(stats) as f, open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as g:
    for row in csv.reader(f):
        row[0] = row[0][:80]
        g.write(",".join(f'"{c}"' for c in row) + "\n")
trace = max(glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
bench_line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")][-1]
bench = json.loads(bench_line)
KERNEL = bench["roofline"]["kernel"]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(trace)) if r["Kernel_Name"] == KERNEL]

# 2. PMC passes
pmc = collections.defaultdict(float)
meta = {}
# (one file per pass: a directory that has collected the files of several runs contributes its newest only)
newest = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    d = os.path.relpath(f, src).split(os.sep)[0]
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in sorted(newest.values()):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] != KERNEL:
            continue
        per[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
        meta = {"VGPR_Count": r["VGPR_Count"], "SGPR_Count": r["SGPR_Count"], "Workgroup_Size": r["Workgroup_Size"], "Grid_Size": r["Grid_Size"]}
    byname = collections.defaultdict(list)
    for (name, _), vals in per.items():
        byname[name].append(sum(vals))
    for name, vals in byname.items():
        pmc[name] = sum(vals) / len(vals)

seg = bench["roofline"]["segments_per_launch"]
cfg = bench["config"]
n_sph = cfg["spheres"]
n_pad = (n_sph + 1) // 2 * 2
sha_file = os.path.join(src, "code_object.sha256")
sha = open(sha_file).read().strip() if os.path.exists(sha_file) else None
lines = []
w = lines.append
w(f"# {tag}: rocprofv3 summary of `python bench.py --workload {cfg['workload'].split(':')[0]}` ({cfg['workload'].split(';')[0]}; 1 x MI355X)")
w(f"code object rt_kernels.hsaco sha256 {sha}")
w(f"kernel-trace: {KERNEL} {len(durs)} dispatches, durations ms {['%.1f' % d for d in durs]}, average {sum(durs) / len(durs):.1f} ms")
w(f"bench.py in the same run: kernel_ms {bench['roofline']['kernel_ms']:.1f} (HIP events in the library), value {bench['value']:.1f} Mrays/s, "
  f"roofline {bench['roofline']['achieved']:.2f} TFLOP/s = {100 * bench['roofline']['frac']:.1f} % of {bench['roofline']['peak']} "
  f"({bench['roofline']['flop_per_segment']} FLOP per segment)")
w(f"dispatch: {meta}")
waves = int(meta.get("Grid_Size", 0)) // 64 if meta else 0
# registers the compiler allotted: from the code object's metadata in the ISA listing (make -C ray_tracing_amd/csrc asm)
vgprs = None
asm = os.path.join(root, "ray_tracing_amd", "csrc", "rt_kernels.s")
if os.path.exists(asm):
    text = open(asm).read()
    m = re.search(r"\.name:\s+" + re.escape(KERNEL) + r"\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text)
    vgprs = int(m.group(1)) if m else None
w(f"  launched: Grid_Size / 64 = {waves} wavefronts.  rocprofv3's VGPR_Count is half of what the compiler reports for the same kernel")
alloc = ((vgprs + 7) // 8 * 8) if vgprs else None
w(f"  (code object metadata: {KERNEL} .vgpr_count {vgprs}, allocated in granules of 8 -> {alloc if alloc else '?'}: "
  f"{(512 // alloc) if alloc else '?'} waves per SIMD of a 512-register file,")
w("  which the measured residency below confirms; the dispatch row divides the descriptor's")
w("  8-register granules by two).")
if pmc.get("SQ_WAVES") and waves:
    w(f"  SQ_WAVES reads {pmc['SQ_WAVES']:.0f} = {pmc['SQ_WAVES'] / waves:.2f} x the wavefronts launched.")
w("")
w("PMC (separate passes, one dispatch each, summed over the chip):")
for k in sorted(pmc):
    w(f"  {k:28s} {pmc[k]:20.0f}")
w("")
w("derived:")
if pmc.get("SQ_INSTS_VALU"):
    w(f"  ray segments per launch            {seg:.0f}")
    w(f"  VALU wave-instructions per segment {pmc['SQ_INSTS_VALU'] / seg * 64:.1f} x 64 (scan alone: {n_pad} x 9 = {n_pad * 9} per bounce segment, {n_pad} x 4 per camera ray)")
    w(f"  active lanes per VALU instruction  {pmc['SQ_THREAD_CYCLES_VALU'] / pmc['SQ_INSTS_VALU']:.1f} of 64 ({100 * pmc['SQ_THREAD_CYCLES_VALU'] / pmc['SQ_INSTS_VALU'] / 64:.1f} %)  <- branch divergence")
    w(f"  LDS instructions per VALU instr    {pmc['SQ_INSTS_LDS'] / pmc['SQ_INSTS_VALU']:.3f}")
    w(f"  SALU instructions per VALU instr   {pmc['SQ_INSTS_SALU'] / pmc['SQ_INSTS_VALU']:.3f}")
if pmc.get("SQ_LDS_IDX_ACTIVE"):
    w(f"  LDS bank-conflict cycles / LDS active cycles  {pmc['SQ_LDS_BANK_CONFLICT'] / pmc['SQ_LDS_IDX_ACTIVE']:.5f}")
if pmc.get("GRBM_GUI_ACTIVE"):
    k_s = sum(durs) / len(durs) * 1e-3
    clk = pmc["GRBM_GUI_ACTIVE"] / 8 / k_s
    w(f"  effective shader clock             {clk / 1e9:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time)")
    w(f"  VALU wave-instructions per SIMD per cycle  {pmc['SQ_INSTS_VALU'] / 1024 / (pmc['GRBM_GUI_ACTIVE'] / 8):.3f} (tools/microbench/vgpr_bank: a pure v_fma_f32 stream issues 0.42 at 8 waves per SIMD, 0.36 at 4)")
    if pmc.get("SQ_WAVE_CYCLES"):
        w(f"  resident waves per SIMD (SQ_WAVE_CYCLES*4 / cycles / 1024)  {pmc['SQ_WAVE_CYCLES'] * 4 / (pmc['GRBM_GUI_ACTIVE'] / 8) / 1024:.2f}  <- occupancy")
if pmc.get("SQ_INSTS_VMEM_RD") is not None and "SQ_INSTS_VMEM_RD" in pmc:
    w(f"  vector-memory wave-instructions per segment x 64: reads {pmc['SQ_INSTS_VMEM_RD'] / seg * 64:.2f}, writes {pmc['SQ_INSTS_VMEM_WR'] / seg * 64:.2f} "
      f"(global + scratch; a spill-free kernel writes only finished pixels), branches {pmc['SQ_INSTS_BRANCH'] / seg * 64:.1f}, "
      f"scalar loads {pmc['SQ_INSTS_SMEM'] / seg * 64:.2f}, MFMA {pmc.get('SQ_INSTS_MFMA', 0):.0f}")
hbm = None
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
    fetch = pmc["FETCH_SIZE"] * 1024 * 2
    write = pmc["WRITE_SIZE"] * 1024
    hbm = fetch + write
    w(f"  HBM traffic per launch             {hbm / 1e6:.1f} MB (FETCH_SIZE {pmc['FETCH_SIZE']:.0f} KiB x2 gfx950 correction + WRITE_SIZE {pmc['WRITE_SIZE']:.0f} KiB);"
      f" algorithmic bytes: {cfg['width']}*{cfg['height']}*4 B frame = {cfg['width'] * cfg['height'] * 4 / 1e6:.1f} MB + {n_pad * 68 / 1e3:.0f} KB scene")
# What the hardware executed, beside the algorithmic count of bench.py's roofline: VALU lane-operations (an upper bound
# of the FLOP actually done is 2 per lane-operation, as if every one were an FMA) and the share of VALU issue slots used
# (a SIMD issues at most one wave64 fp32 instruction every 2 cycles).
executed = None
if pmc.get("SQ_INSTS_VALU") and pmc.get("SQ_THREAD_CYCLES_VALU") and pmc.get("GRBM_GUI_ACTIVE") and durs:
    k_s = sum(durs) / len(durs) * 1e-3
    cycles = pmc["GRBM_GUI_ACTIVE"] / 8
    issue = pmc["SQ_INSTS_VALU"] / 1024 / cycles
    executed = {"valu_wave_instructions": pmc["SQ_INSTS_VALU"], "valu_lane_operations": pmc["SQ_THREAD_CYCLES_VALU"],
                "lane_operations_per_segment": pmc["SQ_THREAD_CYCLES_VALU"] / seg,
                "flop_upper_bound_tflops": 2.0 * pmc["SQ_THREAD_CYCLES_VALU"] / k_s * 1e-12,
                "flop_upper_bound_frac_of_peak": 2.0 * pmc["SQ_THREAD_CYCLES_VALU"] / k_s * 1e-12 / bench["roofline"]["peak"],
                "valu_issue_per_simd_cycle": issue, "valu_issue_frac_of_slots": issue / 0.5,
                "active_lane_frac": pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_INSTS_VALU"] / 64.0,
                "shader_clock_ghz": cycles / k_s * 1e-9,
                "note": "PMC passes of bench.py --steps 1 with this code object; flop_upper_bound counts every VALU lane-operation as an FMA"}
    w(f"  executed VALU lane-operations per segment  {executed['lane_operations_per_segment']:.0f} (algorithmic count: {bench['roofline']['flop_per_segment']} FLOP);"
      f" as if all were FMAs: {executed['flop_upper_bound_tflops']:.1f} TFLOP/s = {100 * executed['flop_upper_bound_frac_of_peak']:.1f} % of peak;"
      f" VALU issue slots used: {100 * executed['valu_issue_frac_of_slots']:.1f} %")
open(os.path.join(dst, f"{tag}_rocprof_summary.txt"), "w").write("\n".join(l for l in lines) + "\n")
if hbm is not None:
    json.dump({"hbm_bytes_per_launch": hbm, "fetch_size_kib": pmc["FETCH_SIZE"], "write_size_kib": pmc["WRITE_SIZE"],
               "code_object_sha256": sha, "workload": cfg["workload"], "executed": executed,
               "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes of bench.py --steps 1; FETCH_SIZE doubled (gfx950 correction)"},
              open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"))
bench["roofline"]["traffic"] = hbm
bench["roofline"]["executed"] = executed
json.dump(bench, open(os.path.join(dst, f"{tag}_bench_1gpu.json"), "w"))
print("\n".join(lines))
