"""ctypes binding of the C-ABI in include/rt_hip.h (ray_tracing_amd/csrc/librt_hip.so).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C ray_tracing_amd/csrc``).
There is no fallback: if the shared object is missing, importing this module raises.
"""
import ctypes as C
import importlib.util
import os

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_CSRC, "librt_hip.so")   # RT_HIP_LIB: use another build of the library

RT_MAT_DIFFUSE, RT_MAT_REFLECT, RT_MAT_DIELECTRIC = 1, 2, 3
RT_FLAG_SAMPLER_GRID, RT_FLAG_QUANT_ROUND, RT_FLAG_BAND_ALBEDO, RT_FLAG_FLIP_SIGNBIT = 1, 2, 4, 8


class RtSphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", C.c_uint32),
                ("albedo", C.c_float * 3), ("param", C.c_float)]


class RtCamera(C.Structure):
    _fields_ = [("aspect_ratio", C.c_float), ("image_width", C.c_uint32), ("samples_per_pixel", C.c_uint32),
                ("max_depth", C.c_uint32), ("vfov", C.c_float), ("lookfrom", C.c_float * 3),
                ("lookat", C.c_float * 3), ("vup", C.c_float * 3), ("image_height", C.c_uint32),
                ("pixel_samples_scale", C.c_float), ("center", C.c_float * 3), ("pixel00_loc", C.c_float * 3),
                ("pixel_delta_u", C.c_float * 3), ("pixel_delta_v", C.c_float * 3), ("u", C.c_float * 3),
                ("v", C.c_float * 3), ("w", C.c_float * 3)]


class RtParams(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_uint32), ("sample2_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
                ("flags", C.c_uint32), ("seed", C.c_uint64), ("t_min", C.c_float), ("t_max", C.c_float),
                ("block_dim", C.c_uint32), ("sample_chunk", C.c_uint32)]


SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", "<u4"),
                         ("albedo", "<f4", 3), ("param", "<f4")])
assert SPHERE_DTYPE.itemsize == C.sizeof(RtSphere) == 36
assert C.sizeof(RtCamera) == 148
assert C.sizeof(RtParams) == 40

# Every symbol include/rt_hip.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_ctx = C.c_void_p
SYMBOLS = {
    "rt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rt_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "rt_destroy": (None, [_ctx]),
    "rt_last_error": (C.c_char_p, [_ctx]),
    "rt_load_module": (C.c_int, [_ctx, _vp, C.c_size_t]),
    "rt_upload_scene": (C.c_int, [_ctx, _vp, C.c_uint32]),
    "rt_render_rows": (C.c_int, [_ctx, C.POINTER(RtCamera), C.POINTER(RtParams), C.c_uint32, C.c_uint32,
                                 _vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    "rt_render_tiles_device": (C.c_int, [_ctx, C.POINTER(RtCamera), C.POINTER(RtParams), C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, _vp, _vp, C.POINTER(C.c_uint64)]),
    "rt_render_tiles_device_static": (C.c_int, [_ctx, C.POINTER(RtCamera), C.POINTER(RtParams), C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_uint32, _vp, _vp, C.POINTER(C.c_uint64)]),
    "rt_comm_id": (C.c_int, [_vp]),
    "rt_comm_init": (C.c_int, [_ctx, _vp, C.c_uint32, C.c_uint32]),
    "rt_render_frame_sharded": (C.c_int, [_ctx, C.POINTER(RtCamera), C.POINTER(RtParams), C.c_uint32, _vp, C.POINTER(C.c_uint64)]),
    "rt_assemble_tiles": (C.c_int, [_ctx, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "rt_primary_hits": (C.c_int, [_ctx, C.POINTER(RtCamera), C.POINTER(RtParams), C.c_uint32, C.c_uint32, _vp]),
    "rt_show_random": (C.c_int, [_ctx, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, _vp]),
    "rt_last_kernel_ms": (C.c_float, [_ctx]),
    "rt_last_kernel_name": (C.c_char_p, [_ctx]),
    "rt_debug_counters": (C.c_int, [_ctx, C.POINTER(C.c_uint64 * 16)]),
    "rt_debug_schedule": (C.c_int, [_ctx, C.c_uint32, C.c_uint32, C.c_int, C.c_int]),
    "rt_debug_fold_workspace_limit": (C.c_int, [_ctx, C.c_uint64]),
    "rt_debug_cull": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "rt_device_info": (C.c_int, [_ctx, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rt_camera_new": (C.c_int, [C.c_float, C.c_uint32, C.POINTER(RtCamera)]),
    "rt_camera_initialize": (C.c_int, [C.POINTER(RtCamera)]),
    "rt_camera_new_cpu": (C.c_int, [C.c_double, C.c_uint32, C.POINTER(RtCamera)]),
    "rt_scene_cpu_default": (C.c_uint32, [_vp, C.c_uint32]),
    "rt_scene_gpu_default": (C.c_uint32, [_vp, C.c_uint32]),
    "rt_scene_book1_final": (C.c_uint32, [C.c_uint64, _vp, C.c_uint32]),
    "rt_scene_grid": (C.c_uint32, [C.c_uint64, C.c_uint32, _vp, C.c_uint32]),
    "rt_scene_mirror_hall": (C.c_uint32, [C.c_uint64, C.c_uint32, _vp, C.c_uint32]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, _vp, C.c_uint32, C.c_uint32]),
    "rt_write_ppm_p3f": (C.c_int, [C.c_char_p, _vp, C.c_uint32, C.c_uint32]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels carry their own libamdhip64.so.7.  Two HIP runtimes in one process cannot both own
    the GPU, so when torch is installed its copy is loaded first (without importing torch); librt_hip.so's
    DT_NEEDED entry has the same soname and binds to it.  Without torch the library uses /opt/rocm/lib."""
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    hip = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(hip):
        C.CDLL(hip, mode=C.RTLD_GLOBAL)
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):                       # the RCCL that goes with that runtime (rt_comm_* load it lazily)
            os.environ.setdefault("RT_RCCL_LIB", rccl)


def lib():
    """The loaded C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ray_tracing_amd/csrc` (there is no CPU fallback)")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
