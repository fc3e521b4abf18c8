"""Randomised GPU-vs-checker stress run, beyond what the test-suite holds: N seeded scenes built to strain the scan
filters (spheres behind rays, nested and overlapping spheres so that bounce rays start inside other spheres, cameras
inside glass, a huge ground, radii over four decades, scenes far from the origin, lists beyond one LDS tile, frames
from 16 to 360 pixels wide), every
sample its own work item under both schedulers plus the fold, each compared bit for bit (linear radiance, RGBA8,
segment count) with the CPU checker (oracle/, test infrastructure).
    python tests/stress_parity.py [N=200] [first_seed=0]
Prints one line per mismatch and a summary; exit code 1 if anything differed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # the repository root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_ffi as oracle
import ray_tracing_amd as rt

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hip = rt.HipWorld(0)
bad = 0
kernels = {}
for case in range(first, first + n_cases):
    rng = np.random.default_rng(9000 + case)
    kind = case % 8
    n = int(rng.choice([3, 9, 40, 130, 500, 1030, 1100])) if kind != 7 else int(rng.choice([1100, 2100]))
    sp = np.zeros(n, rt.SPHERE_DTYPE)
    shift = np.zeros(3)
    if kind == 5:
        shift = rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(2, 4)           # the whole scene far from the origin
    extent = 10.0 ** rng.uniform(-1, 1.5)
    sp["center"] = (rng.normal(0, extent, (n, 3)) + shift).astype(np.float32)
    sp["radius"] = (extent * np.exp(rng.uniform(np.log(0.01), np.log(1.0), n))).astype(np.float32)
    if kind in (1, 2):                                                       # nested: big glass shells around the cloud
        for k in range(min(3, n)):
            sp["center"][k] = (shift + rng.normal(0, 0.1 * extent, 3)).astype(np.float32)
            sp["radius"][k] = np.float32(extent * (1.5 + k))
    if kind == 3:                                                            # a huge ground under everything
        sp["center"][0] = (shift + np.array([0.0, -1000.0 * extent, 0.0])).astype(np.float32)
        sp["radius"][0] = np.float32(1000.0 * extent - 2 * extent)
    sp["material"] = rng.choice([1, 2, 3], n, p=[0.3, 0.4, 0.3])
    if kind in (1, 2):
        sp["material"][:3] = 3
    sp["albedo"] = rng.uniform(0.3, 1.0, (n, 3)).astype(np.float32)
    fuzz = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(0.0, 1.0, n))      # half of the metals are perfect mirrors
    sp["param"] = np.where(sp["material"] == 3, rng.choice([1.5, 1.0 / 1.5, 2.4], n), fuzz).astype(np.float32)
    # (every third frame wide enough for the pixel units' sphere lists to be short, i.e. used)
    width = int(rng.integers(16, 80)) if case % 3 else int(rng.integers(160, 360))
    inside = kind in (2, 4)                                                  # camera inside the cloud / inside a shell
    lookfrom = shift + (rng.normal(0, 0.3 * extent, 3) if inside else rng.normal(0, 1, 3) * extent * rng.uniform(3, 8))
    cam = rt.Camera.look_at(float(rng.choice([16 / 9, 1.0])), width, float(rng.uniform(15, 110)),
                            tuple(lookfrom), tuple(shift + rng.normal(0, 0.2 * extent, 3)))
    spp = int(rng.integers(2, 7)) if case % 3 else int(rng.integers(4, 9))
    depth = int(rng.choice([2, 5, 12, 40]))
    hip.upload_scene(sp)
    ocam = oracle.camera_f32_from_rt(cam.block)
    for chunk, scheds in ((1, ("wave", "pool")), (0, ("auto",))):
        p = rt.DeviceGUI(samples_per_pixel=spp, max_depth=depth, seed=case, sample_chunk=chunk)
        rgba, lin, seg = oracle.render(ocam, sp, p)
        for sched in scheds:
            hip.debug_schedule(scheduler=sched)
            got = hip.render(cam, p, linear=True)
            hip.debug_schedule()
            name = hip.last_kernel_name()
            kernels[name] = kernels.get(name, 0) + 1
            same = (got["rays"] == seg and np.array_equal(got["rgba"], rgba) and
                    np.array_equal(got["linear"].view(np.uint32), lin.view(np.uint32)))
            if not same:
                bad += 1
                diff = int((got["linear"].view(np.uint32) != lin.view(np.uint32)).any(axis=-1).sum())
                print(f"MISMATCH case {case} kind {kind} n {n} {cam.image_width}x{cam.image_height} spp {spp} depth {depth} chunk {chunk} "
                      f"{sched} {name}: {diff} pixels differ, rays {got['rays']} vs {seg}", flush=True)
    if (case - first) % 25 == 24:
        print(f"... {case - first + 1} scenes, {bad} mismatches", flush=True)
print(f"{n_cases} scenes (seeds {first}..{first + n_cases - 1}), kernels run: {kernels}; mismatches: {bad}")
sys.exit(1 if bad else 0)
