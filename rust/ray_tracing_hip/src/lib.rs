//! `HipWorld`: what `CudaWorld` (gpu-implementation/src/ray_tracing.rs:73-211) is to cudarc, over librt_hip.so.
//! The reference's `main.rs` keeps calling `new` / `compile_ptx` / `render`; the frame is the caller's `&mut [u8]`.
pub mod ffi;

use ffi::*;
use std::ffi::CStr;

#[derive(Debug)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "[{}] {}", self.code, self.message)
    }
}
impl std::error::Error for Error {}

/// The knobs of `DeviceGUI` (ray_tracing.rs:43-63) plus the seed that replaces the cuRAND-filled buffer.
#[derive(Clone, Copy, Debug)]
pub struct DeviceGui {
    pub sample2_per_pixel: u32,
    pub block_dim: u32,
    pub max_depth: u32,
    pub show_random: bool,
    pub random_norm: bool,
    pub seed: u64,
}

impl Default for DeviceGui {
    fn default() -> Self {                                   // ray_tracing.rs:53-63
        Self { sample2_per_pixel: 4, block_dim: 1024, max_depth: 4, show_random: false, random_norm: false, seed: 0 }
    }
}

impl DeviceGui {
    pub fn params(&self) -> RtParams {
        RtParams {
            samples_per_pixel: self.sample2_per_pixel * self.sample2_per_pixel,
            sample2_per_pixel: self.sample2_per_pixel,
            max_depth: self.max_depth,
            flags: RT_FLAG_SAMPLER_GRID,                     // ray.cu:52-78
            seed: self.seed,
            t_min: 0.001,
            t_max: 10024.0,                                  // ray.cu:24
            block_dim: self.block_dim / 64 * 64,             // wave64: multiples of 64
            sample_chunk: 1,
        }
    }
}

pub struct HipWorld {
    ctx: *mut rt_ctx,
}

impl HipWorld {
    /// `CudaWorld::new` (ray_tracing.rs:105-127): context on `device`, embedded code object, scene upload.
    pub fn new(device: i32, world: &[RtSphere]) -> Result<Self, Error> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { rt_create(device, &mut ctx) };
        if rc != 0 {
            return Err(Error { code: rc, message: last_error(std::ptr::null()) });
        }
        let me = Self { ctx };
        me.check(unsafe { rt_upload_scene(me.ctx, world.as_ptr(), world.len() as u32) })?;
        Ok(me)
    }

    /// The nine spheres that the reference hard-codes in its kernel (ray.cu:97-106).
    pub fn default_scene() -> Vec<RtSphere> {
        let n = unsafe { rt_scene_gpu_default(std::ptr::null_mut(), 0) };
        let mut v = vec![RtSphere::default(); n as usize];
        unsafe { rt_scene_gpu_default(v.as_mut_ptr(), n) };
        v
    }

    /// `compile_ptx` (ray_tracing.rs:129-161): load a freshly built code object, or the embedded one again.
    pub fn compile(&mut self, hsaco: Option<&[u8]>) -> Result<(), Error> {
        let rc = match hsaco {
            Some(b) => unsafe { rt_load_module(self.ctx, b.as_ptr(), b.len()) },
            None => unsafe { rt_load_module(self.ctx, std::ptr::null(), 0) },
        };
        self.check(rc)
    }

    /// `render` (ray_tracing.rs:163-210): fills `frame` (RGBA8, width * height * 4 bytes); on error the frame is
    /// left untouched, as in :202-204.  Returns the number of ray segments traced.
    pub fn render(&mut self, frame: &mut [u8], cam: &mut RtCamera, gui: &DeviceGui) -> Result<u64, Error> {
        self.check(unsafe { rt_camera_initialize(cam) })?;   // per-frame initialize(), ray_tracing.rs:181
        let need = cam.image_width as usize * cam.image_height as usize * 4;
        if frame.len() < need {
            return Err(Error { code: 10001, message: format!("frame holds {} bytes, {} needed", frame.len(), need) });
        }
        if gui.show_random {                                 // ray.cu:114-118
            let rc = unsafe {
                rt_show_random(self.ctx, cam.image_width, cam.image_height, gui.seed, gui.random_norm as i32, frame.as_mut_ptr())
            };
            return self.check(rc).map(|_| 0);
        }
        let p = gui.params();
        let mut rays = 0u64;
        let rc = unsafe {
            rt_render_rows(self.ctx, cam, &p, 0, cam.image_height, frame.as_mut_ptr(), std::ptr::null_mut(),
                           std::ptr::null_mut(), &mut rays)
        };
        self.check(rc).map(|_| rays)
    }

    /// One frame over several GPUs: every rank (one `HipWorld` per GPU) calls this after `join`; rank 0's frame
    /// receives the result.
    pub fn render_sharded(&mut self, frame: Option<&mut [u8]>, cam: &RtCamera, p: &RtParams) -> Result<u64, Error> {
        let mut rays = 0u64;
        let out = frame.map(|f| f.as_mut_ptr()).unwrap_or(std::ptr::null_mut());
        let rc = unsafe { rt_render_frame_sharded(self.ctx, cam, p, 8, out, &mut rays) };
        self.check(rc).map(|_| rays)
    }

    pub fn comm_id() -> Result<[u8; RT_COMM_ID_BYTES], Error> {
        let mut id = [0u8; RT_COMM_ID_BYTES];
        let rc = unsafe { rt_comm_id(id.as_mut_ptr()) };
        if rc != 0 {
            return Err(Error { code: rc, message: last_error(std::ptr::null()) });
        }
        Ok(id)
    }

    pub fn join(&mut self, id: &[u8; RT_COMM_ID_BYTES], world: u32, rank: u32) -> Result<(), Error> {
        self.check(unsafe { rt_comm_init(self.ctx, id.as_ptr(), world, rank) })
    }

    pub fn last_kernel_ms(&self) -> f32 {
        unsafe { rt_last_kernel_ms(self.ctx) }
    }

    fn check(&self, rc: i32) -> Result<(), Error> {
        if rc == 0 { Ok(()) } else { Err(Error { code: rc, message: last_error(self.ctx) }) }
    }
}

impl Drop for HipWorld {
    fn drop(&mut self) {
        unsafe { rt_destroy(self.ctx) }
    }
}

fn last_error(ctx: *const rt_ctx) -> String {
    let p = unsafe { rt_last_error(ctx) };
    if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() }
}

/// `Camera::new` of the GPU crate (ray_tracing.rs:276-298): defaults, then `initialize()`.
pub fn camera(aspect_ratio: f32, image_width: u32) -> RtCamera {
    let mut c = RtCamera::default();
    unsafe { rt_camera_new(aspect_ratio, image_width, &mut c) };
    c
}
