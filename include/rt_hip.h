/*
 * rt_hip.h -- C-ABI of the MI355X path-tracing library (librt_hip.so).
 *
 * This is the boundary a host in any language binds to (plain pointers and sizes, no HIP or torch
 * types).  It replaces the reference's Rust `CudaWorld` device layer
 * (gpu-implementation/src/ray_tracing.rs:73-211: cudarc context, NVRTC module, kernel launch, D2H copy)
 * with hipModuleLoadData / hipModuleLaunchKernel / hipMemcpy on a gfx950 code object that is built
 * ahead of time and embedded in the library.  Every entry point cites the reference interface it stands for.
 *
 * Conventions: every function returns 0 on success or a non-zero code (a hipError_t value, or
 * RT_ERR_* below); the message is available from rt_last_error().  Nothing aborts or throws across
 * the boundary.  Calls on one context are blocking and must not overlap; use one context per GPU
 * (contexts may be driven from different host threads).  The caller owns every host buffer; the
 * context owns its device buffers and re-allocates them when sizes change.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ERR_INVALID   10001  /* bad argument */
#define RT_ERR_NO_MODULE 10002  /* no code object loaded */
#define RT_ERR_NO_SCENE  10003  /* rt_upload_scene not called yet */
#define RT_ERR_IO        10004  /* file output failed */
#define RT_ERR_COMM      10005  /* RCCL could not be loaded or reported an error */

typedef struct rt_ctx rt_ctx;

/* Number of GPUs this process can use (CudaContext::new(0) assumes one, ray_tracing.rs:107). */
int rt_device_count(int *count);

/* Creates a context on GPU `device`, its stream, and loads the embedded code object.
 * Stands for CudaWorld::new (ray_tracing.rs:106-127); failures are returned, not panics. */
int rt_create(int device, rt_ctx **out);

/* Frees device buffers, module and stream (the reference relies on Drop). */
void rt_destroy(rt_ctx *ctx);

/* Last error message of this context (of the failed rt_create when ctx is NULL).
 * Stands for the error strings the reference pushes to its GUI (ray_tracing.rs:154-158, 202-204). */
const char *rt_last_error(const rt_ctx *ctx);

/* Replaces the kernels by another gfx950 code object image (hipModuleLoadData), or reloads the
 * embedded one when image == NULL.  Stands for CudaWorld::compile_ptx (ray_tracing.rs:129-161);
 * the reference compiles at run time with NVRTC, here the object is built by hipcc --genco.
 * The object must come from this library's rt_kernels.hip with this library's tile size of the streaming
 * kernels (its word rt_stream_tile is checked: the scene's layout in device memory follows from it); any
 * other is refused with RT_ERR_INVALID and the context keeps the module it had. */
int rt_load_module(rt_ctx *ctx, const void *image, size_t size);

/* Uploads the sphere list: AoS host spheres -> device arrays (scan-filter data, exact {c, r*r},
 * shading data {c, radius}, {albedo, fuzz|ior}, material tags).  Stands for the scene the reference builds inside its kernel
 * (gpu-implementation/src/cuda/ray.cu:94-106) and for the `world: &[T]` argument of Camera::render
 * (cpu-implementation/src/lib.rs:221). */
int rt_upload_scene(rt_ctx *ctx, const RtSphere *spheres, uint32_t n);

/* Renders image rows [row0, row0+rows) and copies them to host buffers (blocking).
 *   rgba_out   rows*W*4 bytes RGBA8, row-major                       (required)
 *   linear_out rows*W*4 floats: mean radiance before gamma, r,g,b,1  (may be NULL)
 *   hit_out    rows*W int32: closest sphere of each pixel-centre ray, -1 = none (may be NULL)
 *   rays_out   number of ray segments traced (may be NULL)
 * `cam` must be initialised (rt_camera_initialize or rt_camera_new_cpu).
 * Stands for CudaWorld::render (ray_tracing.rs:163-210) and Camera::render (lib.rs:221-242). */
int rt_render_rows(rt_ctx *ctx, const RtCamera *cam, const RtParams *params, uint32_t row0, uint32_t rows,
                   uint8_t *rgba_out, float *linear_out, int32_t *hit_out, uint64_t *rays_out);

/* Renders an interleaved set of row tiles into caller-owned DEVICE memory (no host copy): tile k
 * (k = 0..n_tiles-1) covers image rows (first_tile + k*tile_stride)*tile_rows ... +tile_rows-1 and is
 * written at local rows k*tile_rows... of the compact output.  Rows past the image end are skipped.
 *   d_rgba   device pointer, n_tiles*tile_rows*W*4 bytes             (required)
 *   d_linear device pointer, n_tiles*tile_rows*W*4 floats            (may be NULL)
 * Blocking: returns after the kernel has finished.  Used to split one image over the GPUs of a node
 * (rank r of n renders tiles r, r+n, ...).  No counterpart in the reference, which is single-GPU
 * (ray_tracing.rs:107); it is the same launch as rt_render_rows with a different row mapping. */
int rt_render_tiles_device(rt_ctx *ctx, const RtCamera *cam, const RtParams *params, uint32_t tile_rows,
                           uint32_t first_tile, uint32_t tile_stride, uint32_t n_tiles,
                           void *d_rgba, void *d_linear, uint64_t *rays_out);

/* rt_render_tiles_device for callers that cannot allocate: code under HIP graph capture set-up, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference (whose CudaWorld allocates in its
 * constructor only, ray_tracing.rs:106-127, and whose kernel needs no device memory beyond its arguments).
 * Same arguments, same results, bit for bit.  The kernels' launch-wide counter block (128 bytes: segments traced,
 * next pixel unit) lives in a statically allocated device array inside the library (256 bytes on every GPU the
 * library runs on, held until the process ends) instead of memory from hipMalloc; this call performs no
 * hipMalloc / hipFree and calls no allocating routine.  The block's size does not depend on the frame, the scene
 * or the sample count; a code object loaded with rt_load_module that states a larger counter block than the
 * array holds is rejected with RT_ERR_INVALID.  All contexts on one GPU share that GPU's array: calls that
 * overlap on one GPU (two contexts driven from two threads) race and must be serialised by the caller.  The
 * array is cleared at the start of every call.  The context, its scene (rt_upload_scene) and the output buffers
 * are created beforehand, where allocation is allowed. */
int rt_render_tiles_device_static(rt_ctx *ctx, const RtCamera *cam, const RtParams *params, uint32_t tile_rows,
                                  uint32_t first_tile, uint32_t tile_stride, uint32_t n_tiles,
                                  void *d_rgba, void *d_linear, uint64_t *rays_out);

/* ---- one frame over the GPUs of a node, gathered to rank 0 with RCCL over xGMI -------------------------
 * One context per GPU and rank (one process per GPU, or one host thread per GPU).  The reference renders
 * on one GPU into a host frame the caller owns (CudaWorld::render, ray_tracing.rs:163-210);
 * rt_render_frame_sharded keeps that shape: every rank calls it with the same camera and parameters, and
 * rank 0's frame_out receives the whole frame.  RCCL is loaded at first use (environment RT_RCCL_LIB names
 * the library; default librccl.so.1), so single-GPU users do not need it. */
#define RT_COMM_ID_BYTES 128
/* Rank 0: a fresh communicator id (ncclGetUniqueId); hand the bytes to the other ranks by any means. */
int rt_comm_id(uint8_t id[RT_COMM_ID_BYTES]);
/* Every rank: join the communicator (ncclCommInitRank on the context's GPU).  Collective. */
int rt_comm_init(rt_ctx *ctx, const uint8_t id[RT_COMM_ID_BYTES], uint32_t world, uint32_t rank);
/* Collective.  The frame is dealt out as interleaved tiles of tile_rows rows (a multiple of 8): rank r renders
 * tiles r, r+world, ...; the tile buffers travel to rank 0 (grouped ncclSend/ncclRecv on the context's
 * stream), a copy kernel puts them in image order, and rank 0's frame_out (H*W*4 bytes, host) gets the RGBA8
 * frame -- bit-identical to rt_render_rows on one GPU.  frame_out is ignored on the other ranks.  rays_out
 * (may be NULL) receives this rank's own segment count.  A context without rt_comm_init is rank 0 of 1.
 * Before any tile travels the ranks exchange one status word (4-byte all-reduce): if a rank failed locally
 * (allocation, launch), it returns its own error and every other rank returns RT_ERR_COMM instead of waiting
 * for it.  A rank on which that exchange itself fails gives up its communicator (ncclCommAbort) and returns the
 * error; the context then refuses further sharded renders (RT_ERR_COMM) -- make a new context and communicator.
 * Arguments must be the same on all ranks. */
int rt_render_frame_sharded(rt_ctx *ctx, const RtCamera *cam, const RtParams *params, uint32_t tile_rows,
                            uint8_t *frame_out, uint64_t *rays_out);
/* The copy kernel of the gather by itself, for hosts that move the tile buffers with a transport of their own
 * (e.g. torch.distributed): d_gathered holds world buffers back to back, each ceil(ceil(H/tile_rows)/world)
 * * tile_rows rows of W RGBA8 pixels as rt_render_tiles_device fills them for rank 0, 1, ...; d_frame receives
 * H*W*4 bytes in image order.  Both are device pointers on the context's GPU.  Blocking. */
int rt_assemble_tiles(rt_ctx *ctx, const void *d_gathered, uint32_t width, uint32_t height, uint32_t tile_rows,
                      uint32_t world, void *d_frame);

/* Closest-hit index map only (no shading).  The reference has no such output; it is the integer
 * result of hit_world (ray.cu:1-18) / `[T]::hit` (lib.rs:127-135) for the first segment. */
int rt_primary_hits(rt_ctx *ctx, const RtCamera *cam, const RtParams *params, uint32_t row0, uint32_t rows,
                    int32_t *hit_out);

/* RNG debug views `show_random` / `random_norm` of the reference kernel (ray.cu:114-118):
 * per pixel three uniform values (norm = 0) or one random unit vector (norm != 0) as RGBA8. */
int rt_show_random(rt_ctx *ctx, uint32_t width, uint32_t height, uint64_t seed, int norm, uint8_t *rgba_out);

/* Milliseconds the last render kernel took on the GPU (hipEvent pair around the launch). */
float rt_last_kernel_ms(const rt_ctx *ctx);
/* Name of the kernel the last render launched (rt_render, rt_render_stream, rt_render_queue, rt_render_pool,
 * rt_render_queue_stream, rt_render_queue_fold, rt_render_pool_fold or rt_render_queue_stream_fold), for profiles. */
const char *rt_last_kernel_name(const rt_ctx *ctx);

/* Diagnostic: the 16 64-bit counters of the last render that asked for rays_out.  out[0] = ray segments;
 * out[8] = pixel units handed out (+ the requests that found none left: one per wavefront, or a few per
 * workgroup on rt_render_pool).  The others are filled only by a code
 * object built with -DRT_STATS: 1..7 loop-trip counts (scan passes, candidates listed, exact-test rounds,
 * decode rounds, unit-vector tries, drain calls, unit-vector rounds), 9..15 shader cycles per kernel section
 * (scan filter of bounce rays, decode, exact tests, hit + scatter, sky / accumulate, hand-out + queue + camera-ray
 * generation, scan filter of camera rays).  No counterpart in the reference. */
int rt_debug_counters(rt_ctx *ctx, uint64_t out[16]);

/* Diagnostic: scheduling experiments on this context.  unit_w x unit_h (powers of two, at most 64 pixels;
 * 0, 0 = the library's choice) forces the shape of the pixel units the wavefronts take when sample_chunk > 0;
 * no_queue != 0 keeps sample_chunk = 1 on the work-item kernel instead of the camera-ray-batch kernels;
 * scheduler picks, for sample_chunk = 1 and 0 on sphere lists that fit one LDS tile, who holds the open pixel units:
 * RT_SCHED_WAVE each wavefront two of its own (rt_render_queue[_fold]), RT_SCHED_POOL the workgroup a pool shared by
 * its wavefronts (rt_render_pool[_fold]), RT_SCHED_AUTO the library's choice by the size of the launch.  None of these
 * changes a single bit of the image (exact accumulation), only the timing.  No counterpart in the reference
 * beyond its block_dim slider (gui.rs:287). */
#define RT_SCHED_AUTO 0
#define RT_SCHED_WAVE 1
#define RT_SCHED_POOL 2
int rt_debug_schedule(rt_ctx *ctx, uint32_t unit_w, uint32_t unit_h, int no_queue, int scheduler);

/* Diagnostic: the largest fold-order workspace this context may allocate, in bytes (0 = the default, 8 GiB).
 * RtParams.sample_chunk = 0 (the reference's summation order) runs on the camera-ray-batch kernels with a workspace of
 * finished samples in device memory, owned by the context and allocated at the first such render:
 *   resident wavefronts x 2 open units x pixels per unit x samples_per_pixel x 16 bytes
 * (4K book scene at 500 spp: 5,120 x 2 x 4 x 500 x 16 B = 328 MB).  A launch whose workspace would exceed the limit,
 * or cannot be allocated, runs on the one-lane-per-pixel kernel instead, which needs none: same image bit for bit,
 * slower.  No counterpart in the reference (its kernel keeps one thread per pixel, ray.cu:109-129). */
int rt_debug_fold_workspace_limit(rt_ctx *ctx, uint64_t bytes);

/* Diagnostic: switches off what lets the camera-ray-batch kernels (sample_chunk = 1 and 0) leave out ray-sphere tests
 * the reference's loop over its sphere list makes (hit_world, ray.cu:1-18; lib.rs:127-135) -- for measurement: with
 * both off every path segment is tested against every sphere of the list, as in the reference; the image is the same
 * bit for bit either way.
 *   unit_lists  != 0 (default): the camera rays of a pixel unit scan the unit's short list of spheres, made once per unit
 *   block_boxes != 0 (default): scenes of 32 spheres and more are kept sorted in space, and a bounce ray's scan skips
 *                               a block of 32 spheres whose bounding box no lane's ray reaches
 * Takes effect at the next launch.  No counterpart in the reference. */
int rt_debug_cull(rt_ctx *ctx, int unit_lists, int block_boxes);

/* Name and compute-unit count of the context's GPU. */
int rt_device_info(const rt_ctx *ctx, char *name, size_t name_size, int *compute_units, int *clock_khz);

/* ---- host-side helpers (no GPU needed) ---- */

/* camera::Camera::new(aspect_ratio, image_width) + initialize() with the reference's defaults
 * (vfov 90, lookfrom (-2,2,1), lookat (0,0,-1), vup (0,1,0); ray_tracing.rs:275-298). */
int rt_camera_new(float aspect_ratio, uint32_t image_width, RtCamera *out);
/* camera::Camera::initialize (ray_tracing.rs:300-336): recomputes the derived fields in f32. */
int rt_camera_initialize(RtCamera *cam);
/* Camera::new(aspect_ratio, image_width, _) of the CPU crate (lib.rs:148-183): fixed pinhole at the
 * origin, computed in f64 and rounded into the f32 block. */
int rt_camera_new_cpu(double aspect_ratio, uint32_t image_width, RtCamera *out);

/* Scene builders.  Each writes up to `cap` spheres and returns the scene's sphere count (call with
 * cap = 0 to size the buffer). */
/* The two spheres of cpu-implementation/src/main.rs:26-28 (Lambertian, albedo 0.5). */
uint32_t rt_scene_cpu_default(RtSphere *out, uint32_t cap);
/* The nine spheres of gpu-implementation/src/cuda/ray.cu:97-106. */
uint32_t rt_scene_gpu_default(RtSphere *out, uint32_t cap);
/* "Ray Tracing in One Weekend" final scene (ground + up to 22x22 small spheres + 3 large ones). */
uint32_t rt_scene_book1_final(uint64_t seed, RtSphere *out, uint32_t cap);
/* Ground + side x side jittered grid of small spheres (side = 128 gives 16,385 spheres). */
uint32_t rt_scene_grid(uint64_t seed, uint32_t side, RtSphere *out, uint32_t cap);
/* Two facing mirrors (ray.cu:103-104) with `n_small` metal/glass spheres between them. */
uint32_t rt_scene_mirror_hall(uint64_t seed, uint32_t n_small, RtSphere *out, uint32_t cap);

/* Binary PPM (P6) of the RGB channels of an RGBA8 frame. */
int rt_write_ppm(const char *path, const uint8_t *rgba, uint32_t width, uint32_t height);
/* The format of the reference's checked-in cpu-implementation/image.ppm: "P3", "W H", "255", then one line per
 * pixel with the three unrounded values 255.999 * linear colour, printed with the shortest digits that round-trip.
 * `linear` is the W*H*4 float buffer rt_render_rows returns. */
int rt_write_ppm_p3f(const char *path, const float *linear, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
