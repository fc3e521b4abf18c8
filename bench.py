#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + bounce ray segments) of the path-tracing hot path on MI355X.

Workload (N = 1 and N > 1 alike): the "Ray Tracing in One Weekend" final scene (485 spheres, mixed
Lambertian / metal / dielectric), 3840x2160, 500 samples per pixel, depth 50 -- the configuration the
target metric is quoted on.  One step = one full frame.  With N GPUs the frame is dealt out as
interleaved 8-row tiles (rank r renders tiles r, r+N, ...; total work fixed = strong scaling) and the
finished tiles are gathered to rank 0 with RCCL inside the timed region.

    python bench.py                      # 1 GPU, defaults
    python bench.py --gpus N             # starts its own N ranks (a child `python -m torch.distributed.run`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.  `value` counts every ray segment all ranks traced in the K timed steps
(in-kernel counters) divided by the max-over-ranks wall time.  `roofline` prices the render kernel
against the FP32 vector peak with the algorithmic FLOP count 18*N_spheres + 64 per segment and the
kernel's own duration (HIP events on the launch stream, inside the library); that count is the algorithm's
(a full ray-sphere test per sphere), not the instructions the kernel issues -- `roofline.executed` carries the
hardware-side figures of the stamped profile under profiles/ when one matches the code object, and `brute_force`
the same frame timed with every segment tested against every sphere (the kernel's culling off).  `e2e_*` is a
separate, untimed-for-`value` leg: the same frame through the blocking host-frame entry point (kernel + D2H, what
the reference's frame time contains).  `cpu_baseline` times the
double-precision CPU restatement of the reference (oracle/, test infrastructure) on a bounded sample of
the same scene on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X vector FP32 peak: 256 CU x 2.4 GHz x 256 FLOP/clk/CU


# name -> (description, width, spp, depth); scene and camera in workload()
WORKLOADS = {
    "c2": ("cpu-implementation default scene (2 spheres, per-column reflectance)", 1920, 100, 50),
    "c3": ("book-1 final scene", 3840, 500, 50),
    "c4": ("16,385-sphere grid scene (sphere list streamed through LDS)", 1920, 64, 16),
    "c5": ("mirror/glass hall, deep bounces", 7680, 256, 100),
}


def workload(rt, args):
    """(spheres, camera, params) of the chosen configuration."""
    _, width, spp, depth = WORKLOADS[args.workload]
    width, spp, depth = args.width or width, args.spp or spp, args.depth or depth
    if args.workload == "c2":
        sp, cam = rt.scene_cpu_default(), rt.Camera.new_cpu(16.0 / 9.0, width)
        p = rt.cpu_params(spp, depth, seed=1, block_dim=args.block_dim, sample_chunk=args.sample_chunk)
        return sp, cam, p
    if args.workload == "c3":
        sp, cam = rt.scene_book1_final(1), rt.Camera.look_at(16.0 / 9.0, width, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0))
    elif args.workload == "c4":
        sp, cam = rt.scene_grid(2, 128), rt.Camera.look_at(16.0 / 9.0, width, 40.0, (0.0, 30.0, 80.0), (0.0, 0.0, 0.0))
    else:
        sp, cam = rt.scene_mirror_hall(3, 64), rt.Camera.look_at(16.0 / 9.0, width, 90.0, (0.0, 0.0, -85.0), (0.0, 0.0, -70.0))
    p = rt.DeviceGUI(samples_per_pixel=spp, max_depth=depth, seed=1, block_dim=args.block_dim, sample_chunk=args.sample_chunk)
    return sp, cam, p


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS),
                    help="c3 (default) is the configuration the metric is quoted on; the others are the remaining "
                         "single-GPU configurations, for profiles of their kernels")
    ap.add_argument("--width", type=int, default=0, help="0 = the workload's own")
    ap.add_argument("--spp", type=int, default=0, help="0 = the workload's own")
    ap.add_argument("--depth", type=int, default=0, help="0 = the workload's own")
    ap.add_argument("--block-dim", type=int, default=0)
    ap.add_argument("--sample-chunk", type=int, default=1,
                    help="samples per work item (RtParams.sample_chunk); 1 = every sample its own item (camera rays in batches); "
                         "0 = the reference's fp32 fold, one lane per pixel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e-steps", type=int, default=2,
                    help="frames of the separate end-to-end leg (kernel + copy of the frame to host memory); 0 = skip")
    ap.add_argument("--brute-steps", type=int, default=1,
                    help="frames of the separate brute-force leg (every segment against every sphere: unit lists and block "
                         "boxes off); 0 = skip")
    ap.add_argument("--cpu-width", type=int, default=1920)
    ap.add_argument("--cpu-spp", type=int, default=32)
    return ap.parse_args()


def cpu_baseline(rt, spheres, depth):
    """The f64 restatement of cpu-implementation/src/lib.rs on this host's cores, bounded sample."""
    import oracle_ffi
    args = ARGS
    cam = rt.Camera.look_at(16.0 / 9.0, args.cpu_width, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0))
    p = rt.DeviceGUI(samples_per_pixel=args.cpu_spp, max_depth=depth, seed=1)
    threads = oracle_ffi.default_threads()
    ocam = oracle_ffi.camera_f64_from_rt(cam.block)
    t0 = time.perf_counter()
    _, _, seg = oracle_ffi.render(ocam, spheres, p, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": seg / dt * 1e-6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"book-1 final scene {cam.image_width}x{cam.image_height}, {args.cpu_spp} spp, depth {depth}, "
                      f"same camera and seed; f64 restatement of lib.rs, {seg} segments in {dt:.1f} s"}


def code_object_sha256():
    import hashlib
    path = os.path.join(ROOT, "ray_tracing_amd", "csrc", "rt_kernels.hsaco")
    return hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None


def stamped_profile(workload):
    """The PMC profile kept under profiles/ -- only one taken with the code object this run executes (SHA-256
    stamp) AND on this workload (the profile records bench.py's own workload string).  Returns (HBM bytes per
    launch, executed-work figures or None)."""
    import glob
    sha = code_object_sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if sha and rec.get("code_object_sha256") == sha and rec.get("workload") == workload:
            return rec.get("hbm_bytes_per_launch"), rec.get("executed")
    return None, None


def spawn_ranks(args):
    """`python bench.py --gpus N` started without a launcher: start the N ranks as a CHILD process (torch's own
    launcher on 127.0.0.1) and exit with its code.  Runs before anything in this process has touched a GPU;
    counting devices does not initialise one.  Rank 0's JSON line reaches stdout through the inherited handle."""
    import socket
    import subprocess
    import torch
    found = torch.cuda.device_count()
    if found < args.gpus:
        print(f"bench.py: needs {args.gpus} GPUs, found {found}", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = ARGS
    import torch
    import torch.distributed as dist
    import ray_tracing_amd as rt
    from ray_tracing_amd import tiles
    TILE_ROWS = tiles.TILE_ROWS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    found = torch.cuda.device_count()
    if local_rank >= found:
        print(f"bench.py: rank {rank} needs GPU {local_rank}; this node has {found} (needs {world} GPUs, found {found})", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    spheres, cam, params = workload(rt, args)
    W, H = cam.image_width, cam.image_height
    args.spp, args.depth, args.width = params.samples_per_pixel, params.max_depth, W
    hip = rt.HipWorld(local_rank, spheres)

    my_tiles = len(tiles.tiles_of_rank(H, rank, world))
    d_rgba = tiles.local_buffer(H, W, world, dev)          # compact per-rank buffer
    # rank 0: every rank's buffer back to back (the gather's destination) and the frame in image order
    gathered = tiles.gathered_buffer(d_rgba, world) if (world > 1 and rank == 0) else None
    d_frame = torch.empty((H, W, 4), dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()                               # buffers zero-filled before the library's stream writes them

    def step():
        """One frame.  Returns (segments, kernel ms, gather ms, assemble ms, frame on rank 0)."""
        rays, ms = hip.render_tiles_device(cam, params, TILE_ROWS, rank, world, my_tiles, d_rgba.data_ptr())   # blocking
        if world == 1:
            return rays, ms, 0.0, 0.0, d_rgba[:H]
        # RCCL gather over xGMI (<= 4.2 MB per peer at 4K) into the contiguous buffer, then the library's copy
        # kernel puts the tiles in image order.  The gather's time on rank 0 includes waiting for the slowest rank.
        ev[0].record()
        tiles.gather_contiguous(d_rgba, gathered, rank, world)
        ev[1].record()
        ev[1].synchronize()                                # d_rgba is rewritten by the next render on another stream
        gather_ms = ev[0].elapsed_time(ev[1])
        assemble_ms = 0.0
        if rank == 0:
            t = time.perf_counter()
            hip.assemble_tiles(gathered.data_ptr(), W, H, TILE_ROWS, world, d_frame.data_ptr())               # blocking
            assemble_ms = (time.perf_counter() - t) * 1e3
        return rays, ms, gather_ms, assemble_ms, d_frame

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays_total, kernel_ms, gather_ms, assemble_ms = 0, [], [], []
    frame = None
    for _ in range(args.steps):
        rays, ms, gms, ams, frame = step()
        rays_total += rays
        kernel_ms.append(ms)
        gather_ms.append(gms)
        assemble_ms.append(ams)
    fence()
    elapsed = time.perf_counter() - t0
    main_kernel = hip.last_kernel_name()

    # End-to-end leg, outside `value`: the frame as the reference's host receives it (CudaWorld::render launches and
    # then copies the frame to host memory, blocking: gpu-implementation/src/ray_tracing.rs:200-208).  One GPU: the
    # library's rt_render_rows into a host frame (kernel + D2H).  N GPUs: render + gather + assembly as above, then
    # rank 0's copy of the assembled frame to host memory.
    e2e_ms, d2h_ms = [], []
    if args.e2e_steps > 0:
        import numpy as np
        host = np.empty((H, W, 4), np.uint8)
        host_t = torch.from_numpy(host)
        for k in range(args.e2e_steps + 1):                # the first one is a warm-up (device frame, page faults)
            fence()
            t = time.perf_counter()
            if world == 1:
                hip.render(cam, params, out=host)
                t_d2h = 0.0
            else:
                _, _, _, _, f = step()
                t1 = time.perf_counter()
                if rank == 0:
                    host_t.copy_(f)
                t_d2h = (time.perf_counter() - t1) * 1e3
            fence()
            if k > 0:
                e2e_ms.append((time.perf_counter() - t) * 1e3)
                d2h_ms.append(t_d2h)

    # Brute-force leg, outside `value` (one GPU only): the same frame with every path segment tested against every
    # sphere, as the reference's loop over its list does -- the pixel units' sphere lists and the blocks' bounding boxes
    # switched off (rt_debug_cull); the frame is the same bit for bit.  This is the figure the algorithmic FLOP count
    # describes literally.
    brute = None
    if world == 1 and args.brute_steps > 0:
        hip.debug_cull(unit_lists=False, block_boxes=False)
        b_ms, b_rays = [], 0
        for k in range(args.brute_steps + 1):              # the first one is a warm-up
            r, ms_k, _, _, _ = step()
            if k > 0:
                b_ms.append(ms_k)
                b_rays = r
        brute = (sum(b_ms) / len(b_ms), b_rays, hip.last_kernel_name())
        hip.debug_cull()

    mean = lambda v: sum(v) / max(1, len(v))
    stats = torch.tensor([elapsed, float(rays_total), mean(kernel_ms), mean(gather_ms)], dtype=torch.float64, device=dev)
    per_rank = None
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, rays_all, kms_max = tmax[0].item(), tsum[1].item(), tmax[2].item()
        every = [torch.empty_like(stats) for _ in range(world)]
        dist.all_gather(every, stats)
        per_rank = {"wall_s": [e[0].item() for e in every], "segments": [e[1].item() for e in every],
                    "kernel_ms": [e[2].item() for e in every], "gather_ms": [e[3].item() for e in every]}
    else:
        elapsed_max, rays_all, kms_max = stats[0].item(), stats[1].item(), stats[2].item()

    if rank == 0:
        n_sph = len(spheres)
        flop_per_segment = 18 * n_sph + 64
        # dominant kernel = the render kernel on this rank: its own segments over its own average duration
        rank0_rays_per_launch = rays_total / max(1, args.steps)
        k_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        achieved = rank0_rays_per_launch * flop_per_segment / (k_ms * 1e-3) * 1e-12 if k_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes kept under profiles/ -- only if they were taken with the code
        # object this run executes (the file is stamped with its SHA-256) and on this workload; else null
        workload_text = (f"{args.workload}: {WORKLOADS[args.workload][0]}, {n_sph} spheres, {W}x{H}, {args.spp} spp, max depth {args.depth}, "
                         f"jittered sampler, seed 1, sample_chunk {args.sample_chunk}; frame split into interleaved {TILE_ROWS}-row tiles over "
                         f"{world} GPU(s), RCCL gather to rank 0")
        traffic, executed = stamped_profile(workload_text) if world == 1 else (None, None)
        out = {
            "metric": "Mrays/s (primary+bounce) at 1/2/4/8 MI355X + % of FP32 VALU peak",
            "value": rays_all / elapsed_max * 1e-6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / max(1, args.steps) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_text,
                       "spheres": n_sph, "width": W, "height": H, "spp": args.spp, "max_depth": args.depth, "sample_chunk": args.sample_chunk,
                       "segments_per_frame": rays_all / max(1, args.steps), "parallelism": f"rowtiles{world}"},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": main_kernel, "kernel_ms": k_ms, "flop_per_segment": flop_per_segment,
                         "segments_per_launch": rank0_rays_per_launch,
                         # what `achieved` counts: the work of the reference's loop over the whole sphere list (one full
                         # ray-sphere test to the discriminant, 18 FLOP, per sphere and segment + 64 per segment), not
                         # the instructions issued.  The kernel reaches the same hits without most of those tests --
                         # camera rays scan their pixel unit's short sphere list, bounce rays skip the blocks of 32
                         # spheres whose bounding box they miss and run a 9-operation filter on the rest, the full test
                         # on the few candidates -- so the figure is a throughput in the metric's own unit and CAN EXCEED
                         # the peak; the hardware-side numbers are in `executed`.
                         "count": "FLOP of the reference's brute-force loop (18 per ray-sphere test x spheres + 64 per segment) per second, "
                                  "not executed FLOP: the kernel skips tests the reference makes, frac can exceed 1; see executed",
                         # hardware-side figures of the PMC profile taken with this code object on this workload
                         # (profiles/), or null: VALU lane-operations actually executed, and VALU issue slots used
                         "executed": executed},
            # the same frame without the tests' culling (see above): kernel, its time, Mrays/s and the algorithmic
            # FLOP/s it then sustains, which here are FLOP the kernel's filter + exact test stand for one to one
            "brute_force": None if brute is None else {
                "kernel": brute[2], "kernel_ms": brute[0], "value": brute[1] / (brute[0] * 1e-3) * 1e-6, "unit": "Mrays/s",
                "achieved": brute[1] * flop_per_segment / (brute[0] * 1e-3) * 1e-12,
                "frac": brute[1] * flop_per_segment / (brute[0] * 1e-3) * 1e-12 / FP32_VECTOR_PEAK_TFLOPS,
                "same_frame": brute[1] == rank0_rays_per_launch},
            "kernel_ms_max_over_ranks": kms_max,
            # where a multi-GPU step goes: each rank's own render kernel, its gather (on rank 0 this includes the
            # wait for the slowest rank, i.e. load imbalance and launch skew), rank 0's copy kernel
            "per_rank": per_rank,
            "rank0_gather_ms": mean(gather_ms), "rank0_assemble_ms": mean(assemble_ms),
            # end-to-end leg (not part of `value`): frame in host memory; None when skipped
            "e2e_ms_per_step": mean(e2e_ms) if e2e_ms else None,
            "e2e_value": (rays_all / max(1, args.steps)) / (mean(e2e_ms) * 1e-3) * 1e-6 if e2e_ms else None,
            "e2e_rank0_d2h_ms": mean(d2h_ms) if (e2e_ms and world > 1) else None,
            "e2e_note": ("rt_render_rows: kernel + blocking copy of the RGBA8 frame to pageable host memory" if world == 1 else
                         "render + RCCL gather + assembly + rank 0's copy of the frame to pageable host memory") if e2e_ms else None,
            "checksum": int(frame.to(torch.int64).sum().item()) if frame is not None else None,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "c3":
            out["cpu_baseline"] = cpu_baseline(rt, spheres, args.depth)
        print(json.dumps(out), flush=True)
    hip.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    ARGS = parse()
    if ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ and "LOCAL_RANK" not in os.environ:
        sys.exit(spawn_ranks(ARGS))
    main()
