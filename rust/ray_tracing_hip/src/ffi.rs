//! Raw declarations of include/rt_hip.h and include/rt_types.h.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int};

#[repr(C)]
pub struct rt_ctx {
    _private: [u8; 0],
}

/// include/rt_types.h RtSphere (36 bytes); stands for `Sphere` + `Material` of library.cu:152-196.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtSphere {
    pub center: [f32; 3],
    pub radius: f32,
    pub material: u32,
    pub albedo: [f32; 3],
    pub param: f32,
}

/// include/rt_types.h RtCamera (148 bytes): the `#[repr(C)]` camera of gpu-implementation/src/ray_tracing.rs:251-272.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtCamera {
    pub aspect_ratio: f32,
    pub image_width: u32,
    pub samples_per_pixel: u32,
    pub max_depth: u32,
    pub vfov: f32,
    pub lookfrom: [f32; 3],
    pub lookat: [f32; 3],
    pub vup: [f32; 3],
    pub image_height: u32,
    pub pixel_samples_scale: f32,
    pub center: [f32; 3],
    pub pixel00_loc: [f32; 3],
    pub pixel_delta_u: [f32; 3],
    pub pixel_delta_v: [f32; 3],
    pub u: [f32; 3],
    pub v: [f32; 3],
    pub w: [f32; 3],
}

/// include/rt_types.h RtParams (40 bytes).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RtParams {
    pub samples_per_pixel: u32,
    pub sample2_per_pixel: u32,
    pub max_depth: u32,
    pub flags: u32,
    pub seed: u64,
    pub t_min: f32,
    pub t_max: f32,
    pub block_dim: u32,
    pub sample_chunk: u32,
}

pub const RT_MAT_DIFFUSE: u32 = 1;
pub const RT_MAT_REFLECT: u32 = 2;
pub const RT_MAT_DIELECTRIC: u32 = 3;
pub const RT_FLAG_SAMPLER_GRID: u32 = 1;
pub const RT_FLAG_QUANT_ROUND: u32 = 2;
pub const RT_FLAG_BAND_ALBEDO: u32 = 4;
pub const RT_FLAG_FLIP_SIGNBIT: u32 = 8;
pub const RT_COMM_ID_BYTES: usize = 128;

extern "C" {
    pub fn rt_device_count(count: *mut c_int) -> c_int;
    pub fn rt_create(device: c_int, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_destroy(ctx: *mut rt_ctx);
    pub fn rt_last_error(ctx: *const rt_ctx) -> *const c_char;
    pub fn rt_load_module(ctx: *mut rt_ctx, image: *const u8, size: usize) -> c_int;
    pub fn rt_upload_scene(ctx: *mut rt_ctx, spheres: *const RtSphere, n: u32) -> c_int;
    pub fn rt_render_rows(ctx: *mut rt_ctx, cam: *const RtCamera, params: *const RtParams, row0: u32, rows: u32,
                          rgba_out: *mut u8, linear_out: *mut f32, hit_out: *mut i32, rays_out: *mut u64) -> c_int;
    pub fn rt_render_tiles_device(ctx: *mut rt_ctx, cam: *const RtCamera, params: *const RtParams, tile_rows: u32,
                                  first_tile: u32, tile_stride: u32, n_tiles: u32, d_rgba: *mut u8, d_linear: *mut f32,
                                  rays_out: *mut u64) -> c_int;
    /// The same without any allocation: counter block in the library's static device array.
    pub fn rt_render_tiles_device_static(ctx: *mut rt_ctx, cam: *const RtCamera, params: *const RtParams, tile_rows: u32,
                                         first_tile: u32, tile_stride: u32, n_tiles: u32, d_rgba: *mut u8,
                                         d_linear: *mut f32, rays_out: *mut u64) -> c_int;
    pub fn rt_primary_hits(ctx: *mut rt_ctx, cam: *const RtCamera, params: *const RtParams, row0: u32, rows: u32,
                           hit_out: *mut i32) -> c_int;
    pub fn rt_show_random(ctx: *mut rt_ctx, width: u32, height: u32, seed: u64, norm: c_int, rgba_out: *mut u8) -> c_int;
    pub fn rt_last_kernel_ms(ctx: *const rt_ctx) -> f32;
    /// Largest fold-order workspace (sample_chunk = 0) the context may allocate; 0 = the default of 8 GiB.
    pub fn rt_debug_fold_workspace_limit(ctx: *mut rt_ctx, bytes: u64) -> c_int;
    pub fn rt_debug_cull(ctx: *mut rt_ctx, unit_lists: c_int, block_boxes: c_int) -> c_int;
    pub fn rt_comm_id(id: *mut u8) -> c_int;
    pub fn rt_comm_init(ctx: *mut rt_ctx, id: *const u8, world: u32, rank: u32) -> c_int;
    pub fn rt_render_frame_sharded(ctx: *mut rt_ctx, cam: *const RtCamera, params: *const RtParams, tile_rows: u32,
                                   frame_out: *mut u8, rays_out: *mut u64) -> c_int;
    pub fn rt_camera_new(aspect_ratio: f32, image_width: u32, out: *mut RtCamera) -> c_int;
    pub fn rt_camera_initialize(cam: *mut RtCamera) -> c_int;
    pub fn rt_camera_new_cpu(aspect_ratio: f64, image_width: u32, out: *mut RtCamera) -> c_int;
    pub fn rt_scene_cpu_default(out: *mut RtSphere, cap: u32) -> u32;
    pub fn rt_scene_gpu_default(out: *mut RtSphere, cap: u32) -> u32;
    pub fn rt_scene_book1_final(seed: u64, out: *mut RtSphere, cap: u32) -> u32;
    pub fn rt_write_ppm(path: *const c_char, rgba: *const u8, width: u32, height: u32) -> c_int;
}
