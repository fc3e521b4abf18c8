/*
 * rt_oracle.h -- CPU restatement of the reference path tracer.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the checker the HIP path is compared against.  Nothing in the product
 * (the C-ABI render library, the host mirror, the Python package) links, loads or calls it;
 * only tests/, the smoke check and the bench's cpu_baseline leg do.
 *
 * One source (rt_oracle_impl.h) is compiled twice:
 *   or64_*  double precision, no fused multiply-add: follows cpu-implementation/src/lib.rs
 *           (cgmath Vector3<f64> arithmetic).
 *   or32_*  single precision with explicit fmaf in a fixed order: follows the op order of
 *           gpu-implementation/src/cuda/{floatN_helper,library,ray}.cu as nvcc contracts it;
 *           this is the arithmetic the HIP kernel reproduces bit for bit.
 * Both are driven by the same counter-based RNG (PCG32 XSH-RR, library.cu:7-16, keyed on
 * pixel, sample and bounce) because the reference's own streams are not reproducible
 * (thread_rng on the CPU, a racy global on the GPU).
 *
 * Pinned by the reference's only known-answer data, cpu-implementation/image.ppm: camera rays
 * and sky shading bit-exactly, the sphere silhouette exactly, hit point/normal to 1e-12
 * (tests/test_oracle_golden.py).  Scatter, RNG use, gamma/quantisation have no reference
 * vectors: parity unpinned there, they follow the cited lines.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include "../include/rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Camera in the precision of the path it feeds. */
typedef struct OrCameraD {
    uint32_t image_width, image_height;
    double center[3], pixel00_loc[3], pixel_delta_u[3], pixel_delta_v[3];
} OrCameraD;

typedef struct OrCameraF {
    uint32_t image_width, image_height;
    float center[3], pixel00_loc[3], pixel_delta_u[3], pixel_delta_v[3];
} OrCameraF;

/* ---- RNG (shared by both precisions) ---- */
/* One PCG32 XSH-RR step on (state, inc); returns the 32-bit output (library.cu:7-16). */
uint32_t or_pcg32_next(uint64_t *state, uint64_t inc);
/* pcg32_srandom_r(initstate, initseq) of the PCG reference implementation. */
void or_pcg32_seed(uint64_t *state, uint64_t *inc, uint64_t initstate, uint64_t initseq);
/* Stream for (seed, pixel, sample, k): k = 0 feeds the camera jitter, k = j+1 the scatter that
 * ends path segment j. */
void or_stream_init(uint64_t *state, uint64_t *inc, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k);
/* u32 -> [0,1): (u >> 8) * 2^-24. */
float or_u01(uint32_t u);

/* ---- cameras ---- */
/* Camera::new(aspect_ratio, image_width, _) of lib.rs:148-183, in f64. */
void or64_camera_new(double aspect_ratio, uint32_t image_width, OrCameraD *out);
/* camera::Camera::initialize of ray_tracing.rs:300-336, in f32, on the 148-byte block. */
void or32_camera_initialize(RtCamera *cam);
void or32_camera_from_rt(const RtCamera *cam, OrCameraF *out);
void or64_camera_from_rt(const RtCamera *cam, OrCameraD *out);   /* widening, for f64 renders of look-at cameras */
void or32_camera_from_d(const OrCameraD *cam, OrCameraF *out);   /* rounding, for f32 renders of the lib.rs camera */

/* ---- pieces, exposed for known-answer tests ---- */
/* Ray through the centre of pixel (x,y): sub-pixel offset 0. */
void or64_pixel_center_ray(const OrCameraD *cam, uint32_t x, uint32_t y, double orig[3], double dir[3]);
void or32_pixel_center_ray(const OrCameraF *cam, uint32_t x, uint32_t y, float orig[3], float dir[3]);
/* Miss shading: lib.rs:203-205 with blend = reflectance (f64) / ray.cu:43-45 (f32, blend 0.5). */
void or64_miss_color(const double dir[3], double blend, double rgb[3]);
void or32_miss_color(const float dir[3], float blend, float rgb[3]);

/* One scatter event (Material::scatter library.cu:131-149,175-183 + the book's dielectric): new direction for
 * incoming direction d at a surface with normal n (facing the ray; front = the ray comes from outside).
 * `param` is fuzz (clamped to <= 1 as library.cu:144 does) or the index of refraction. */
void or64_scatter_dir(uint32_t material, double param, const double d[3], const double n[3], int front, uint64_t seed,
                      uint32_t pixel, uint32_t sample, uint32_t k, uint32_t flags, double out[3]);
void or32_scatter_dir(uint32_t material, float param, const float d[3], const float n[3], int front, uint64_t seed,
                      uint32_t pixel, uint32_t sample, uint32_t k, uint32_t flags, float out[3]);
/* The random unit vector of stream (seed, pixel, sample, k): lib.rs:31-47. */
void or64_unit_vector(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k, double out[3]);
void or32_unit_vector(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k, float out[3]);

/* Closest hit of every pixel-centre ray: sphere index or -1; optional t, point(3), normal(3) per pixel
 * (Sphere::hit lib.rs:96-125 / library.cu:198-227 under the closest-hit fold lib.rs:127-135 / ray.cu:1-18). */
void or64_primary_hits(const OrCameraD *cam, const RtSphere *spheres, uint32_t n, double t_min, double t_max,
                       uint32_t flags, uint32_t row0, uint32_t rows, int threads,
                       int32_t *idx, double *t, double *point, double *normal);
void or32_primary_hits(const OrCameraF *cam, const RtSphere *spheres, uint32_t n, float t_min, float t_max,
                       uint32_t flags, uint32_t row0, uint32_t rows, int threads,
                       int32_t *idx, float *t, float *point, float *normal);

/* Full render of rows [row0, row0+rows): RGBA8 (4 B/pixel), optional linear radiance (mean over samples,
 * before gamma; 4 values per pixel: r,g,b,1) and optional total ray-segment count.
 * Camera::render lib.rs:221-242 / kernel `render` ray.cu:92-130. */
void or64_render(const OrCameraD *cam, const RtSphere *spheres, uint32_t n, const RtParams *params,
                 uint32_t row0, uint32_t rows, int threads,
                 uint8_t *rgba, double *linear, uint64_t *segments);
void or32_render(const OrCameraF *cam, const RtSphere *spheres, uint32_t n, const RtParams *params,
                 uint32_t row0, uint32_t rows, int threads,
                 uint8_t *rgba, float *linear, uint64_t *segments);

/* The same, with an optional per-pixel signature (rows*W words): a hash of the sphere indices the pixel's paths
 * hit, segment by segment (-1 = sky).  Two renders of one pixel in different precisions followed the same paths
 * exactly when their signatures agree. */
void or64_render_sig(const OrCameraD *cam, const RtSphere *spheres, uint32_t n, const RtParams *params,
                     uint32_t row0, uint32_t rows, int threads,
                     uint8_t *rgba, double *linear, uint64_t *segments, uint64_t *path_sig);
void or32_render_sig(const OrCameraF *cam, const RtSphere *spheres, uint32_t n, const RtParams *params,
                     uint32_t row0, uint32_t rows, int threads,
                     uint8_t *rgba, float *linear, uint64_t *segments, uint64_t *path_sig);

/* RNG debug views of ray.cu:114-118: per pixel, 3 uniform floats (norm = 0) or a random unit vector (norm = 1),
 * from stream (seed, pixel, 0, 0), quantised by to_pixel (ray.cu:81-89). */
void or32_show_random(uint32_t width, uint32_t height, uint64_t seed, int norm, uint8_t *rgba);

#ifdef __cplusplus
}
#endif
#endif
