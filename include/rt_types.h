/*
 * rt_types.h -- plain-old-data types shared by the C-ABI render library (rt_hip.h),
 * the host-side C++ mirror of the reference API (rt_host.hpp) and the CPU checker
 * under oracle/.  Pure C, no HIP or torch types.
 *
 * Reference citations are relative to the upstream repository root.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Material tags.  DIFFUSE/REFLECT keep the numbering of the reference's
 * `enum class MaterialType { DIFFUSE=1, REFLECT }` (gpu-implementation/src/cuda/library.cu:152-155).
 * DIELECTRIC has no counterpart in the reference; it follows "Ray Tracing in One Weekend"
 * (Snell refraction + Schlick reflectance). */
enum {
    RT_MAT_DIFFUSE    = 1,
    RT_MAT_REFLECT    = 2,
    RT_MAT_DIELECTRIC = 3
};

/* Host-side sphere, array-of-structs, 36 bytes -- same shape as the reference's device
 * `Sphere{float3 center; float radius; Material{type; union{Diffuse{color}, Reflect{color,fuzz}}}}`
 * (library.cu:125-196).  `param` is fuzz (REFLECT, clamped to <=1 like library.cu:144)
 * or the index of refraction (DIELECTRIC); unused for DIFFUSE. */
typedef struct RtSphere {
    float    center[3];
    float    radius;
    uint32_t material;
    float    albedo[3];
    float    param;
} RtSphere;

/* Camera block, 148 bytes, passed to the kernel by value.  Field order and offsets are those of
 * the reference's `#[repr(C)] camera::Camera` (gpu-implementation/src/ray_tracing.rs:251-272)
 * and its device mirror (library.cu:54-75). */
typedef struct RtCamera {
    float    aspect_ratio;        /*   0 */
    uint32_t image_width;         /*   4 */
    uint32_t samples_per_pixel;   /*   8 */
    uint32_t max_depth;           /*  12 */
    float    vfov;                /*  16 */
    float    lookfrom[3];         /*  20 */
    float    lookat[3];           /*  32 */
    float    vup[3];              /*  44 */
    uint32_t image_height;        /*  56 */
    float    pixel_samples_scale; /*  60 */
    float    center[3];           /*  64 */
    float    pixel00_loc[3];      /*  76 */
    float    pixel_delta_u[3];    /*  88 */
    float    pixel_delta_v[3];    /* 100 */
    float    u[3];                /* 112 */
    float    v[3];                /* 124 */
    float    w[3];                /* 136 */
} RtCamera;

/* Render flags: select between the two behaviours the reference has for the same step. */
enum {
    /* Sub-pixel sample positions on a deterministic n x n grid, n = sample2_per_pixel
     * (ray.cu:52-78).  Clear: two uniform draws in [-0.5,0.5) per sample (lib.rs:208-219). */
    RT_FLAG_SAMPLER_GRID = 1u << 0,
    /* 8-bit quantiser `(sqrt(x)*255).round() as u8` (lib.rs:236).  Clear: truncation of
     * clamp(sqrt(x)*255, 0, 255) (ray.cu:81-89,129). */
    RT_FLAG_QUANT_ROUND  = 1u << 1,
    /* cpu-implementation shading: every sphere is Lambertian with the per-column scalar
     * reflectance 0.1 + 0.2*(x / (W/5)), which is also the sky blend factor, applied in
     * recursion order, and no near-zero guard on the scatter direction
     * (lib.rs:194-205, 225-226). */
    RT_FLAG_BAND_ALBEDO  = 1u << 2,
    /* Normal flip rule `!is_sign_negative(n.d)` (lib.rs:49-58).  Clear: `n.d > 0`
     * (floatN_helper.cu:129-136). */
    RT_FLAG_FLIP_SIGNBIT = 1u << 3
};

/* Per-frame knobs -- the headless counterpart of the reference's `DeviceGUI`
 * (ray_tracing.rs:43-63) plus the constants its two binaries hard-code. */
typedef struct RtParams {
    uint32_t samples_per_pixel;  /* jitter sampler: samples per pixel; grid sampler: must equal sample2_per_pixel^2 */
    uint32_t sample2_per_pixel;  /* grid sampler: n of the n x n grid (ray.cu:124); ignored otherwise */
    uint32_t max_depth;          /* a path has at most max_depth segments (lib.rs:190-192, ray.cu:29-32) */
    uint32_t flags;              /* RT_FLAG_* */
    uint64_t seed;               /* render seed of the counter-based RNG */
    float    t_min;              /* 0.001 in both reference paths (lib.rs:193, ray.cu:24) */
    float    t_max;              /* +inf on the CPU path (lib.rs:193), 10024 on the GPU path (ray.cu:24) */
    uint32_t block_dim;          /* threads per workgroup, multiple of 64 in 64..1024 (ray_tracing.rs:59,190-192); 0 = library default */
    /* Accumulation of the samples of a pixel.
     * 0: all samples are added in order in fp32, the reference's fold (lib.rs:228-231, ray.cu:123-129).  With two or
     *    more samples per pixel this runs on the camera-ray-batch kernels too, which park finished samples in a device
     *    workspace owned by the context (size and limit: rt_debug_fold_workspace_limit in rt_hip.h); without room for
     *    it, on the one-lane-per-pixel kernel -- the same bits either way.
     * C > 0 (and C < samples_per_pixel): runs of C consecutive samples are added in order in fp32; the run sums are
     *    then added exactly (each converted to 2^-32 fixed point, 64-bit integer addition) and the mean is rounded
     *    once.  Exact addition is order-independent, so the runs of a pixel can be traced by any lane at any time:
     *    this is what lets the kernel hand samples to lanes dynamically.  Differs from the fold by rounding only
     *    (about 1e-7 relative).  C = 1 (every sample its own item) is the fastest setting when the scene fits one
     *    LDS tile: camera rays are then traced 64 at a time with a cheaper scan and bounce rays come from a queue. */
    uint32_t sample_chunk;
} RtParams;

#ifdef __cplusplus
}
#endif

#endif /* RT_TYPES_H */
