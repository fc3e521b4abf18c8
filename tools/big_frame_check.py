"""Index-width check: a 16384 x 9216 frame (151 M pixels) on the GPU, rows of it against the CPU checker."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ray_tracing_amd as rt, oracle_ffi
sp = rt.scene_book1_final(1)
cam = rt.Camera.look_at(16 / 9, 16384, 20.0, (13, 2, 3), (0, 0, 0))
p = rt.DeviceGUI(samples_per_pixel=2, max_depth=8, seed=3, sample_chunk=1)
w = rt.HipWorld(0, sp)
r = w.render(cam, p)
print(cam.image_width, cam.image_height, r["rays"], r["kernel_ms"], w.last_kernel_name())
row0 = cam.image_height - 3
rgba, lin, seg = oracle_ffi.render(oracle_ffi.camera_f32_from_rt(cam.block), sp, p, row0=row0, rows=3)
print("last rows equal:", np.array_equal(r["rgba"][row0:], rgba))
rgba, lin, seg = oracle_ffi.render(oracle_ffi.camera_f32_from_rt(cam.block), sp, p, row0=5000, rows=2)
print("rows 5000 equal:", np.array_equal(r["rgba"][5000:5002], rgba))
