"""MI355X-native path tracer: the per-pixel hot path of NitroSniper/ray_tracing behind its own
scene / camera / render interface.  See include/rt_hip.h for the C-ABI this package binds."""
from .api import (Camera, DeviceGUI, Dielectric, Diffuse, HipWorld, Reflect, RtError, Sphere, cpu_params,
                  scene_book1_final, scene_cpu_default, scene_gpu_default, scene_grid, scene_mirror_hall, world,
                  write_ppm, write_ppm_p3f)
from ._ffi import (RT_FLAG_BAND_ALBEDO, RT_FLAG_FLIP_SIGNBIT, RT_FLAG_QUANT_ROUND, RT_FLAG_SAMPLER_GRID,
                   RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE, RT_MAT_REFLECT, RtCamera, RtParams, SPHERE_DTYPE)

__all__ = [n for n in dir() if not n.startswith("_")]
