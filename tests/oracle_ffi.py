"""ctypes binding of the CPU checker (oracle/librt_oracle.so).

Test infrastructure: imported by tests/, the smoke check and the bench's cpu_baseline leg only.
The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librt_oracle.so")


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


class OrCameraD(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("center", C.c_double * 3),
                ("pixel00_loc", C.c_double * 3), ("pixel_delta_u", C.c_double * 3), ("pixel_delta_v", C.c_double * 3)]


class OrCameraF(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("center", C.c_float * 3),
                ("pixel00_loc", C.c_float * 3), ("pixel_delta_u", C.c_float * 3), ("pixel_delta_v", C.c_float * 3)]


SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", "<u4"),
                         ("albedo", "<f4", 3), ("param", "<f4")])
assert SPHERE_DTYPE.itemsize == C.sizeof(RtSphere) == 36
assert C.sizeof(RtCamera) == 148

FLAG_SAMPLER_GRID = 1
FLAG_QUANT_ROUND = 2
FLAG_BAND_ALBEDO = 4
FLAG_FLIP_SIGNBIT = 8

_lib = None


def build(force=False):
    """Compile the checker with the recipe under oracle/ (gcc only)."""
    if force or not os.path.exists(ORACLE_SO) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(ORACLE_SO)
            for f in ("rt_oracle.c", "rt_oracle_impl.h", "rt_oracle.h")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        L.or_pcg32_next.restype = C.c_uint32
        L.or_pcg32_next.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        L.or_pcg32_seed.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64]
        L.or_stream_init.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32,
                                     C.c_uint32, C.c_uint32]
        L.or_u01.restype = C.c_float
        L.or_u01.argtypes = [C.c_uint32]
        L.or64_camera_new.argtypes = [C.c_double, C.c_uint32, C.POINTER(OrCameraD)]
        L.or32_camera_initialize.argtypes = [C.c_void_p]
        L.or32_camera_from_rt.argtypes = [C.c_void_p, C.POINTER(OrCameraF)]
        L.or64_camera_from_rt.argtypes = [C.c_void_p, C.POINTER(OrCameraD)]
        L.or32_camera_from_d.argtypes = [C.POINTER(OrCameraD), C.POINTER(OrCameraF)]
        vp = C.c_void_p
        L.or64_pixel_center_ray.argtypes = [C.POINTER(OrCameraD), C.c_uint32, C.c_uint32, vp, vp]
        L.or32_pixel_center_ray.argtypes = [C.POINTER(OrCameraF), C.c_uint32, C.c_uint32, vp, vp]
        L.or64_miss_color.argtypes = [vp, C.c_double, vp]
        L.or32_miss_color.argtypes = [vp, C.c_float, vp]
        L.or64_primary_hits.argtypes = [C.POINTER(OrCameraD), vp, C.c_uint32, C.c_double, C.c_double, C.c_uint32,
                                        C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, vp]
        L.or32_primary_hits.argtypes = [C.POINTER(OrCameraF), vp, C.c_uint32, C.c_float, C.c_float, C.c_uint32,
                                        C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, vp]
        L.or64_render.argtypes = [C.POINTER(OrCameraD), vp, C.c_uint32, vp, C.c_uint32, C.c_uint32,
                                  C.c_int, vp, vp, C.POINTER(C.c_uint64)]
        L.or32_render.argtypes = [C.POINTER(OrCameraF), vp, C.c_uint32, vp, C.c_uint32, C.c_uint32,
                                  C.c_int, vp, vp, C.POINTER(C.c_uint64)]
        L.or64_scatter_dir.argtypes = [C.c_uint32, C.c_double, vp, vp, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, vp]
        L.or32_scatter_dir.argtypes = [C.c_uint32, C.c_float, vp, vp, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, vp]
        L.or64_unit_vector.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.or32_unit_vector.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.or64_render_sig.argtypes = L.or64_render.argtypes + [vp]
        L.or32_render_sig.argtypes = L.or32_render.argtypes + [vp]
        L.or32_show_random.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, vp]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def as_spheres(spheres):
    a = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    return a


def camera_new_f64(aspect, width):
    cam = OrCameraD()
    lib().or64_camera_new(float(aspect), int(width), C.byref(cam))
    return cam


def camera_f32_from_d(camd):
    cam = OrCameraF()
    lib().or32_camera_from_d(C.byref(camd), C.byref(cam))
    return cam


def camera_f32_from_rt(rt):
    """`rt` is any ctypes structure with the 148-byte RtCamera layout (the product's or this module's)."""
    assert C.sizeof(rt) == 148
    cam = OrCameraF()
    lib().or32_camera_from_rt(C.addressof(rt), C.byref(cam))
    return cam


def camera_f64_from_rt(rt):
    cam = OrCameraD()
    lib().or64_camera_from_rt(C.addressof(rt), C.byref(cam))
    return cam


def primary_hits(cam, spheres, t_min=0.001, t_max=float("inf"), flags=0, row0=0, rows=None, threads=None,
                 details=False):
    """Closest-hit index map of pixel-centre rays; precision follows the camera type."""
    f64 = isinstance(cam, OrCameraD)
    W, H = cam.image_width, cam.image_height
    rows = H - row0 if rows is None else rows
    sp = as_spheres(spheres)
    idx = np.empty((rows, W), np.int32)
    ft = np.float64 if f64 else np.float32
    t = point = normal = None
    if details:
        t = np.empty((rows, W), ft)
        point = np.empty((rows, W, 3), ft)
        normal = np.empty((rows, W, 3), ft)
    fn = lib().or64_primary_hits if f64 else lib().or32_primary_hits
    fn(C.byref(cam), _ptr(sp), len(sp), t_min, t_max, flags, row0, rows, threads or default_threads(),
       _ptr(idx), _ptr(t), _ptr(point), _ptr(normal))
    return (idx, t, point, normal) if details else idx


def render(cam, spheres, params, row0=0, rows=None, threads=None, signatures=False):
    """Full render; returns (rgba u8 [rows,W,4], linear [rows,W,4], segments[, path signatures u64 [rows,W]])."""
    f64 = isinstance(cam, OrCameraD)
    W, H = cam.image_width, cam.image_height
    rows = H - row0 if rows is None else rows
    sp = as_spheres(spheres)
    rgba = np.empty((rows, W, 4), np.uint8)
    lin = np.empty((rows, W, 4), np.float64 if f64 else np.float32)
    seg = C.c_uint64(0)
    if signatures:
        sig = np.empty((rows, W), np.uint64)
        fn = lib().or64_render_sig if f64 else lib().or32_render_sig
        fn(C.byref(cam), _ptr(sp), len(sp), C.addressof(params), row0, rows, threads or default_threads(),
           _ptr(rgba), _ptr(lin), C.byref(seg), _ptr(sig))
        return rgba, lin, seg.value, sig
    fn = lib().or64_render if f64 else lib().or32_render
    fn(C.byref(cam), _ptr(sp), len(sp), C.addressof(params), row0, rows, threads or default_threads(),
       _ptr(rgba), _ptr(lin), C.byref(seg))
    return rgba, lin, seg.value


def scatter_dir(material, param, d, n, front=True, key=(0, 0, 0, 1), flags=0, f64=True):
    """New direction of one scatter event; key = (seed, pixel, sample, k) of the stream it may draw from."""
    ft = np.float64 if f64 else np.float32
    dv, nv, out = np.asarray(d, ft).copy(), np.asarray(n, ft).copy(), np.zeros(3, ft)
    fn = lib().or64_scatter_dir if f64 else lib().or32_scatter_dir
    fn(material, param, _ptr(dv), _ptr(nv), int(front), key[0], key[1], key[2], key[3], flags, _ptr(out))
    return out


def unit_vector(key, f64=True):
    out = np.zeros(3, np.float64 if f64 else np.float32)
    (lib().or64_unit_vector if f64 else lib().or32_unit_vector)(key[0], key[1], key[2], key[3], _ptr(out))
    return out


def show_random(width, height, seed, norm):
    out = np.empty((height, width, 4), np.uint8)
    lib().or32_show_random(width, height, seed, int(norm), _ptr(out))
    return out
