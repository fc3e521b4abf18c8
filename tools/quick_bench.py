"""Quick single-GPU throughput probe: book-1 final scene, kernel time from the library's HIP events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt

width = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 50
block = int(sys.argv[4]) if len(sys.argv) > 4 else 0
scene = sys.argv[5] if len(sys.argv) > 5 else "book1"
chunk = int(os.environ.get("RT_CHUNK", "0"))
if scene == "book1":
    sp = rt.scene_book1_final(1); cam = rt.Camera.look_at(16 / 9, width, 20.0, (13, 2, 3), (0, 0, 0))
elif scene == "grid":
    sp = rt.scene_grid(2, 128); cam = rt.Camera.look_at(16 / 9, width, 40.0, (0, 30, 80), (0, 0, 0))
elif scene == "hall":
    sp = rt.scene_mirror_hall(3, 64); cam = rt.Camera.look_at(16 / 9, width, 90.0, (0, 0, -85), (0, 0, -70))
else:
    sp = rt.scene_cpu_default(); cam = rt.Camera.new_cpu(16 / 9, width)
p = rt.cpu_params(spp, depth, block_dim=block, sample_chunk=chunk) if scene == "cpu" else rt.DeviceGUI(samples_per_pixel=spp, max_depth=depth, seed=1, block_dim=block, sample_chunk=chunk)
w = rt.HipWorld(0, sp)
# scheduling experiments, from the environment of the A/B scripts: RT_UNIT=WxH, RT_NO_QUEUE=1, RT_SCHED=wave|pool
_u = os.environ.get("RT_UNIT")
w.debug_schedule(unit=tuple(int(v) for v in _u.split("x")) if _u else None, no_queue=os.environ.get("RT_NO_QUEUE") == "1", scheduler=os.environ.get("RT_SCHED", "auto"))
if os.environ.get("RT_BRUTE") == "1":                      # every segment against every sphere (rt_debug_cull)
    w.debug_cull(unit_lists=False, block_boxes=False)
hs = os.environ.get("RT_HSACO")
if hs:
    w.load_module(open(hs, "rb").read())
print(w.device_info(), hs)


def sections(c):
    names = ["scan", "decode", "exact", "hit+scatter", "sky/add", "hand-out/queue/raygen", "camera-ray filter"]
    tot = sum(c[9:16]) or 1
    return ", ".join(f"{n} {100 * v / tot:.1f}% ({v / 1e9:.2f} G)" for n, v in zip(names, c[9:16]))

n = len(sp)
for it in range(3):
    t0 = time.time()
    r = w.render(cam, p)
    dt = time.time() - t0
    ms = r["kernel_ms"]
    import zlib
    crc = zlib.crc32(r["rgba"].tobytes()) if "rgba" in r else 0
    flops = r["rays"] * (18 * n + 64)
    print(f"{scene} {cam.image_width}x{cam.image_height} spp={spp} depth={depth} n={n}: rays={r['rays']} "
          f"({r['rays'] / (cam.image_width * cam.image_height * spp):.2f}/sample) kernel {ms:.2f} ms wall {dt * 1e3:.1f} ms "
          + (f"stats {w.debug_counters()[:9]} [15]={w.debug_counters()[15]} cycle shares {sections(w.debug_counters())} " if hs and ("stats" in hs or "diag" in hs) else "") +
          f"crc {crc:08x} {w.last_kernel_name()} -> {r['rays'] / ms * 1e-3:.1f} Mrays/s, {flops / ms * 1e-9:.2f} TFLOP/s ({flops / ms * 1e-9 / 157.3 * 100:.1f}% of 157.3)", flush=True)
