"""HBM write bytes of the render kernel against the pixel-unit shape (run under `rocprofv3 --pmc WRITE_SIZE`): the
mirror hall at 1920x1080, 64 spp, one launch per unit shape, wavefront scheduler, then the pooled one."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt  # noqa: E402

sp = rt.scene_mirror_hall(3, 64)
cam = rt.Camera.look_at(16.0 / 9.0, 1920, 90.0, (0.0, 0.0, -85.0), (0.0, 0.0, -70.0))
p = rt.DeviceGUI(samples_per_pixel=64, max_depth=100, seed=1, sample_chunk=1)
with rt.HipWorld(0, sp) as hip:
    buf = torch.empty((cam.image_height, cam.image_width), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for sched in ("wave", "pool"):
        for unit in ((4, 1), (8, 1), (16, 1), (16, 2), (16, 4)):
            hip.debug_schedule(unit=unit, scheduler=sched)
            r, ms = hip.render_tiles_device(cam, p, cam.image_height, 0, 1, 1, buf.data_ptr())
            print(sched, unit, hip.last_kernel_name(), f"{ms:.2f} ms", flush=True)
