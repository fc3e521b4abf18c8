"""Single-GPU estimate of multi-GPU strong scaling: times the render kernel of rank r of `world` (interleaved
8-row tiles of the 4K book-1 frame, or RT_SCENE=grid|hall: the other benchmark frames) for several r, and reports T1 / (world * max_r T_r)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import tiles

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 500
worlds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
scene = os.environ.get("RT_SCENE", "book1")          # book1 (4K, depth 50), grid (1080p, depth 16), hall (8K, depth 100)
if scene == "grid":
    sp = rt.scene_grid(2, 128); cam = rt.Camera.look_at(16 / 9, 1920, 40.0, (0, 30, 80), (0, 0, 0)); depth = 16
elif scene == "hall":
    sp = rt.scene_mirror_hall(3, 64); cam = rt.Camera.look_at(16 / 9, 7680, 90.0, (0, 0, -85), (0, 0, -70)); depth = 100
else:
    sp = rt.scene_book1_final(1); cam = rt.Camera.look_at(16 / 9, 3840, 20.0, (13, 2, 3), (0, 0, 0)); depth = 50
p = rt.DeviceGUI(samples_per_pixel=spp, max_depth=depth, seed=1, sample_chunk=int(os.environ.get('RT_CHUNK', '0')))
w = rt.HipWorld(0, sp)
# scheduling experiments, from the environment of the A/B scripts: RT_UNIT=WxH, RT_NO_QUEUE=1, RT_SCHED=wave|pool
_u = os.environ.get("RT_UNIT")
w.debug_schedule(unit=tuple(int(v) for v in _u.split("x")) if _u else None, no_queue=os.environ.get("RT_NO_QUEUE") == "1", scheduler=os.environ.get("RT_SCHED", "auto"))
hs = os.environ.get("RT_HSACO")
if hs:
    w.load_module(open(hs, "rb").read())
H, W = cam.image_height, cam.image_width
t1 = None
for world in worlds:
    buf = tiles.local_buffer(H, W, world, "cuda")
    torch.cuda.synchronize()
    res = []
    for r in sorted(set([0, world // 2, world - 1])):
        n = len(tiles.tiles_of_rank(H, r, world))
        w.render_tiles_device(cam, p, tiles.TILE_ROWS, r, world, n, buf.data_ptr())
        rays, ms = w.render_tiles_device(cam, p, tiles.TILE_ROWS, r, world, n, buf.data_ptr())
        res.append((r, rays, ms))
    tmax = max(ms for _, _, ms in res)
    if world == 1:
        t1 = tmax
    eff = t1 / (world * tmax) if t1 else float("nan")
    print(f"world {world}: " + ", ".join(f"rank {r}: {rays / 1e6:.0f} Mrays {ms:.1f} ms" for r, rays, ms in res) +
          f" -> est. efficiency {eff * 100:.1f} %", flush=True)
