"""Host-side mirror of the reference's scene / camera / render interface, over the C-ABI.

Names follow the reference: ``Camera`` (gpu-implementation/src/ray_tracing.rs:253-337 and
cpu-implementation/src/lib.rs:137-183), ``Sphere`` (lib.rs:85-94, library.cu:186-196), the material
constructors ``Diffuse`` / ``Reflect`` (library.cu:125-150) plus ``Dielectric``, ``DeviceGUI``
(ray_tracing.rs:43-63) and ``HipWorld`` standing for ``CudaWorld`` (ray_tracing.rs:73-211).
All computation happens in librt_hip.so; this module only marshals.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import (RT_FLAG_BAND_ALBEDO, RT_FLAG_FLIP_SIGNBIT, RT_FLAG_QUANT_ROUND, RT_FLAG_SAMPLER_GRID,
                   RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE, RT_MAT_REFLECT, RtCamera, RtParams, SPHERE_DTYPE)


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


# ---------------------------------------------------------------- materials and spheres
def Diffuse(color):
    """Diffuse(color) -- library.cu:125-136."""
    return (RT_MAT_DIFFUSE, tuple(float(c) for c in color), 0.0)


def Reflect(color, fuzz):
    """Reflect(color, fuzz), fuzz clamped to <= 1 -- library.cu:138-150."""
    return (RT_MAT_REFLECT, tuple(float(c) for c in color), min(float(fuzz), 1.0))


def Dielectric(ior, color=(1.0, 1.0, 1.0)):
    """Dielectric(index of refraction); not in the reference (book section 11)."""
    return (RT_MAT_DIELECTRIC, tuple(float(c) for c in color), float(ior))


def Sphere(center, radius, material=None):
    """Sphere::new(center, radius) (lib.rs:91) / Sphere(center, radius, material) (library.cu:193)."""
    mat = material if material is not None else Diffuse((0.5, 0.5, 0.5))
    s = np.zeros((), SPHERE_DTYPE)
    s["center"] = center
    s["radius"] = radius
    s["material"] = mat[0]
    s["albedo"] = mat[1]
    s["param"] = mat[2]
    return s


def world(spheres):
    """A sphere list (`&[Sphere]`, lib.rs:221) as one contiguous array."""
    if isinstance(spheres, np.ndarray) and spheres.dtype == SPHERE_DTYPE:
        return np.ascontiguousarray(spheres)
    return np.array([np.asarray(s, SPHERE_DTYPE) for s in spheres], dtype=SPHERE_DTYPE)


def _scene(fn, *args):
    n = fn(*args, None, 0)
    out = np.zeros(n, SPHERE_DTYPE)
    fn(*args, _ffi.ptr(out), n)
    return out


def scene_cpu_default():
    return _scene(_ffi.lib().rt_scene_cpu_default)


def scene_gpu_default():
    return _scene(_ffi.lib().rt_scene_gpu_default)


def scene_book1_final(seed=1):
    return _scene(_ffi.lib().rt_scene_book1_final, seed)


def scene_grid(seed=2, side=128):
    return _scene(_ffi.lib().rt_scene_grid, seed, side)


def scene_mirror_hall(seed=3, n_small=64):
    return _scene(_ffi.lib().rt_scene_mirror_hall, seed, n_small)


# ---------------------------------------------------------------- camera
class Camera:
    """The 148-byte camera block and its two constructors."""

    def __init__(self, block):
        self.block = block

    @classmethod
    def new(cls, aspect_ratio, image_width):
        """camera::Camera::new(aspect_ratio, image_width) -- ray_tracing.rs:275-298."""
        b = RtCamera()
        rc = _ffi.lib().rt_camera_new(aspect_ratio, image_width, C.byref(b))
        if rc:
            raise RtError(rc, "rt_camera_new: bad arguments")
        return cls(b)

    @classmethod
    def new_cpu(cls, aspect_ratio, image_width):
        """Camera::new(aspect_ratio, image_width, _) of the CPU crate -- lib.rs:148-183."""
        b = RtCamera()
        rc = _ffi.lib().rt_camera_new_cpu(aspect_ratio, image_width, C.byref(b))
        if rc:
            raise RtError(rc, "rt_camera_new_cpu: bad arguments")
        return cls(b)

    @classmethod
    def look_at(cls, aspect_ratio, image_width, vfov, lookfrom, lookat, vup=(0.0, 1.0, 0.0)):
        cam = cls.new(aspect_ratio, image_width)
        cam.block.vfov = vfov
        cam.block.lookfrom[:] = lookfrom
        cam.block.lookat[:] = lookat
        cam.block.vup[:] = vup
        cam.initialize()
        return cam

    def initialize(self):
        """camera::Camera::initialize -- ray_tracing.rs:300-336."""
        rc = _ffi.lib().rt_camera_initialize(C.byref(self.block))
        if rc:
            raise RtError(rc, "rt_camera_initialize: bad camera")

    @property
    def image_width(self):
        return self.block.image_width

    @property
    def image_height(self):
        return self.block.image_height


def DeviceGUI(sample2_per_pixel=4, block_dim=0, max_depth=4, samples_per_pixel=None, flags=None, seed=0,
              t_min=0.001, t_max=10024.0, sample_chunk=0):
    """Per-frame knobs; the defaults are the reference's DeviceGUI::default (ray_tracing.rs:53-63) and the
    GPU path's t range (ray.cu:24).  With samples_per_pixel given the jitter sampler is used instead of the
    n x n grid.  sample_chunk: see RtParams in include/rt_types.h (0 = the reference's fp32 fold)."""
    p = RtParams()
    if samples_per_pixel is None:
        p.samples_per_pixel = sample2_per_pixel * sample2_per_pixel
        p.sample2_per_pixel = sample2_per_pixel
        p.flags = RT_FLAG_SAMPLER_GRID if flags is None else flags
    else:
        p.samples_per_pixel = samples_per_pixel
        p.sample2_per_pixel = 1
        p.flags = 0 if flags is None else flags
    p.max_depth = max_depth
    p.seed = seed
    p.t_min = t_min
    p.t_max = t_max
    p.block_dim = block_dim
    p.sample_chunk = sample_chunk
    return p


def cpu_params(samples_per_pixel=64, max_depth=20, seed=0, block_dim=0, sample_chunk=0):
    """The CPU crate's behaviour (lib.rs:184-242 with main.rs:16-19 defaults): jittered samples, per-column
    reflectance bands, sign-bit flip rule, rounding quantiser, t in (0.001, inf)."""
    return DeviceGUI(samples_per_pixel=samples_per_pixel, max_depth=max_depth, seed=seed, block_dim=block_dim,
                     flags=RT_FLAG_BAND_ALBEDO | RT_FLAG_FLIP_SIGNBIT | RT_FLAG_QUANT_ROUND,
                     t_min=0.001, t_max=float("inf"), sample_chunk=sample_chunk)


# ---------------------------------------------------------------- device world
class HipWorld:
    """Stands for CudaWorld (ray_tracing.rs:73-211): owns the GPU context, the uploaded scene and the
    device frame; `render` is blocking and returns host arrays."""

    def __init__(self, device=0, spheres=None):
        self._lib = _ffi.lib()
        self._ctx = C.c_void_p()
        rc = self._lib.rt_create(device, C.byref(self._ctx))
        if rc:
            msg = self._lib.rt_last_error(None)
            raise RtError(rc, msg.decode() if msg else "rt_create failed")
        self.device = device
        if spheres is not None:
            self.upload_scene(spheres)

    def _check(self, rc):
        if rc:
            msg = self._lib.rt_last_error(self._ctx)
            raise RtError(rc, msg.decode() if msg else "error")

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.rt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def load_module(self, image=None):
        """compile_ptx counterpart: (re)load a gfx950 code object, default the embedded one."""
        if image is None:
            self._check(self._lib.rt_load_module(self._ctx, None, 0))
        else:
            buf = C.create_string_buffer(bytes(image), len(image))
            self._check(self._lib.rt_load_module(self._ctx, C.cast(buf, C.c_void_p), len(image)))

    def upload_scene(self, spheres):
        w = world(spheres)
        self._check(self._lib.rt_upload_scene(self._ctx, _ffi.ptr(w), len(w)))
        self.n_spheres = len(w)

    def render(self, camera, params, row0=0, rows=None, linear=False, hits=False, out=None):
        """Returns dict(rgba[rows,W,4] u8, rays, kernel_ms[, linear[rows,W,4] f32][, hits[rows,W] i32]).
        out: a caller-owned contiguous [rows,W,4] u8 array to receive the frame (`frame: &mut [u8]`,
        ray_tracing.rs:163) instead of a fresh one."""
        W, H = camera.image_width, camera.image_height
        rows = H - row0 if rows is None else rows
        if out is not None and (out.dtype != np.uint8 or out.shape != (rows, W, 4) or not out.flags["C_CONTIGUOUS"]):
            raise ValueError("out must be a C-contiguous uint8 array of shape (rows, W, 4)")
        rgba = out if out is not None else np.empty((rows, W, 4), np.uint8)
        lin = np.empty((rows, W, 4), np.float32) if linear else None
        hit = np.empty((rows, W), np.int32) if hits else None
        rays = C.c_uint64(0)
        self._check(self._lib.rt_render_rows(self._ctx, C.byref(camera.block), C.byref(params), row0, rows,
                                             _ffi.ptr(rgba), _ffi.ptr(lin), _ffi.ptr(hit), C.byref(rays)))
        out = {"rgba": rgba, "rays": rays.value, "kernel_ms": self._lib.rt_last_kernel_ms(self._ctx)}
        if linear:
            out["linear"] = lin
        if hits:
            out["hits"] = hit
        return out

    def render_tiles_device(self, camera, params, tile_rows, first_tile, tile_stride, n_tiles, d_rgba, d_linear=None,
                            static_scratch=False):
        """Interleaved row tiles into caller-owned device memory (integer device addresses).  static_scratch: the
        allocation-free entry point (counter block in the library's static device array)."""
        rays = C.c_uint64(0)
        fn = self._lib.rt_render_tiles_device_static if static_scratch else self._lib.rt_render_tiles_device
        self._check(fn(self._ctx, C.byref(camera.block), C.byref(params), tile_rows,
                                                     first_tile, tile_stride, n_tiles, C.c_void_p(d_rgba),
                                                     C.c_void_p(d_linear) if d_linear else None, C.byref(rays)))
        return rays.value, self._lib.rt_last_kernel_ms(self._ctx)

    def assemble_tiles(self, d_gathered, width, height, tile_rows, world, d_frame):
        """Gathered tile buffers of `world` ranks (device address) -> frame in image order (device address)."""
        self._check(self._lib.rt_assemble_tiles(self._ctx, C.c_void_p(d_gathered), width, height, tile_rows, world,
                                                C.c_void_p(d_frame)))

    @staticmethod
    def device_count():
        n = C.c_int(0)
        rc = _ffi.lib().rt_device_count(C.byref(n))
        if rc != 0:
            raise RtError(rc, (_ffi.lib().rt_last_error(None) or b"").decode())
        return n.value

    @staticmethod
    def comm_id():
        """A fresh RCCL communicator id (128 bytes); rank 0 creates it and hands it to the other ranks."""
        buf = (C.c_uint8 * 128)()
        rc = _ffi.lib().rt_comm_id(C.byref(buf))
        if rc != 0:
            raise RtError(rc, (_ffi.lib().rt_last_error(None) or b"").decode())
        return bytes(buf)

    def comm_init(self, comm_id, world, rank):
        """Join the communicator as rank `rank` of `world` (collective over the ranks' contexts)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(comm_id)
        self._check(self._lib.rt_comm_init(self._ctx, C.byref(buf), world, rank))

    def render_frame_sharded(self, camera, params, tile_rows=8):
        """Collective: this rank renders its interleaved row tiles; rank 0 returns the assembled RGBA8 frame
        (the other ranks return None for it).  A context that never joined a communicator is rank 0 of 1."""
        W, H = camera.image_width, camera.image_height
        frame = np.empty((H, W, 4), np.uint8)
        rays = C.c_uint64(0)
        self._check(self._lib.rt_render_frame_sharded(self._ctx, C.byref(camera.block), C.byref(params), tile_rows,
                                                      _ffi.ptr(frame), C.byref(rays)))
        return {"rgba": frame, "rays": rays.value, "kernel_ms": self._lib.rt_last_kernel_ms(self._ctx)}

    def primary_hits(self, camera, params, row0=0, rows=None):
        W, H = camera.image_width, camera.image_height
        rows = H - row0 if rows is None else rows
        hit = np.empty((rows, W), np.int32)
        self._check(self._lib.rt_primary_hits(self._ctx, C.byref(camera.block), C.byref(params), row0, rows,
                                              _ffi.ptr(hit)))
        return hit

    def show_random(self, width, height, seed=0, norm=False):
        out = np.empty((height, width, 4), np.uint8)
        self._check(self._lib.rt_show_random(self._ctx, width, height, seed, int(norm), _ffi.ptr(out)))
        return out

    def last_kernel_name(self):
        return (self._lib.rt_last_kernel_name(self._ctx) or b"").decode()

    def debug_counters(self):
        out = (C.c_uint64 * 16)()
        self._check(self._lib.rt_debug_counters(self._ctx, C.byref(out)))
        return list(out)

    def debug_schedule(self, unit=None, no_queue=False, scheduler="auto"):
        """Scheduling experiments: unit = (w, h) forces the pixel-unit shape (None: the library's choice);
        no_queue keeps sample_chunk = 1 on the work-item kernel; scheduler = "wave" | "pool" | "auto" picks who holds
        the open pixel units on the camera-ray-batch kernel (each wavefront its own, or a pool per workgroup).
        The image does not change."""
        w, h = unit if unit else (0, 0)
        self._check(self._lib.rt_debug_schedule(self._ctx, w, h, int(bool(no_queue)),
                                                {"auto": 0, "wave": 1, "pool": 2}[scheduler]))

    def debug_fold_workspace_limit(self, nbytes=0):
        """Largest fold-order workspace (sample_chunk = 0 on the camera-ray-batch kernels) the context may allocate;
        larger launches run on the one-lane-per-pixel kernel.  0 = the default (8 GiB)."""
        self._check(self._lib.rt_debug_fold_workspace_limit(self._ctx, nbytes))

    def debug_cull(self, unit_lists=True, block_boxes=True):
        """Measurement aid: with both off every path segment is tested against every sphere of the list, as in the
        reference's loop (the image does not change).  unit_lists: camera rays scan their pixel unit's short sphere list;
        block_boxes: bounce rays skip blocks of 32 spheres whose bounding box they miss (scenes of 32 spheres and more)."""
        self._check(self._lib.rt_debug_cull(self._ctx, int(bool(unit_lists)), int(bool(block_boxes))))

    def device_info(self):
        name = C.create_string_buffer(600)
        cus, khz = C.c_int(0), C.c_int(0)
        self._check(self._lib.rt_device_info(self._ctx, name, 600, C.byref(cus), C.byref(khz)))
        return name.value.decode(), cus.value, khz.value


def write_ppm(path, rgba):
    a = np.ascontiguousarray(rgba, np.uint8)
    rc = _ffi.lib().rt_write_ppm(str(path).encode(), _ffi.ptr(a), a.shape[1], a.shape[0])
    if rc:
        raise RtError(rc, f"rt_write_ppm({path}) failed")


def write_ppm_p3f(path, linear):
    """The reference's image.ppm format: P3 header, one line of three unrounded 255.999*linear values per pixel."""
    a = np.ascontiguousarray(linear, np.float32)
    rc = _ffi.lib().rt_write_ppm_p3f(str(path).encode(), _ffi.ptr(a), a.shape[1], a.shape[0])
    if rc:
        raise RtError(rc, f"rt_write_ppm_p3f({path}) failed")
