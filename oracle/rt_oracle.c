/*
 * rt_oracle.c -- CPU restatement of the reference path tracer.  TEST INFRASTRUCTURE ONLY:
 * see rt_oracle.h for what it follows, how it is pinned and who may call it.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rt_oracle.h"

/* ------------------------------------------------------------------ RNG */

/* PCG32 XSH-RR, 64-bit LCG state, multiplier 6364136223846793005 (library.cu:7-16).  The reference
 * uses inc|1 with inc = 0; here inc selects the per-pixel stream as in the PCG reference code. */
uint32_t or_pcg32_next(uint64_t *state, uint64_t inc) {
    uint64_t old = *state;
    *state = old * 6364136223846793005ULL + (inc | 1u);
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}

void or_pcg32_seed(uint64_t *state, uint64_t *inc, uint64_t initstate, uint64_t initseq) {
    *state = 0u;
    *inc = (initseq << 1u) | 1u;
    (void)or_pcg32_next(state, *inc);
    *state += initstate;
    (void)or_pcg32_next(state, *inc);
}

static inline uint64_t or_mix64(uint64_t z) { /* splitmix64 finaliser */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void or_stream_init(uint64_t *state, uint64_t *inc, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k) {
    uint64_t key = or_mix64(seed ^ (((uint64_t)sample << 32) | (uint64_t)k));
    or_pcg32_seed(state, inc, key, (uint64_t)pixel);
}

/* The reference maps with ldexpf(u, -32) (library.cu:18-24), which can round up to exactly 1.0;
 * the top 24 bits give an exact value in [0,1). */
float or_u01(uint32_t u) { return (float)(u >> 8) * 0x1p-24f; }

/* ------------------------------------------------------------------ cameras */

void or64_camera_new(double aspect_ratio, uint32_t image_width, OrCameraD *out) {
    uint32_t height = (uint32_t)((double)image_width / aspect_ratio); /* lib.rs:149-156 */
    if (height == 0) height = 1;
    double f_width = (double)image_width, f_height = (double)height;
    double vp_height = 2.0;
    double vp_width = vp_height * aspect_ratio;
    double focal_length = 1.0;
    double center[3] = {0.0, 0.0, 0.0};
    double vp[3] = {vp_width, -vp_height, 0.0};
    double div[3] = {f_width, f_height, 1.0};
    double unit_z[3] = {0.0, 0.0, 1.0};
    double pixel_delta[3];
    for (int k = 0; k < 3; k++) pixel_delta[k] = vp[k] / div[k]; /* lib.rs:167-168 */
    out->image_width = image_width;
    out->image_height = height;
    for (int k = 0; k < 3; k++) {
        double top_left = center[k] - unit_z[k] * focal_length - vp[k] / 2.0; /* lib.rs:171 */
        out->center[k] = center[k];
        out->pixel00_loc[k] = top_left + pixel_delta[k] / 2.0;               /* lib.rs:172 */
        out->pixel_delta_u[k] = 0.0;
        out->pixel_delta_v[k] = 0.0;
    }
    /* lib.rs multiplies (x, y, 0) element-wise by pixel_delta: x only meets delta.x, y only delta.y. */
    out->pixel_delta_u[0] = pixel_delta[0];
    out->pixel_delta_v[1] = pixel_delta[1];
}

static float f_dot(const float *u, const float *v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; } /* ray_tracing.rs:218-220 */
static void f_cross(const float *u, const float *v, float *o) { /* ray_tracing.rs:222-228 */
    o[0] = u[1] * v[2] - u[2] * v[1];
    o[1] = u[2] * v[0] - u[0] * v[2];
    o[2] = u[0] * v[1] - u[1] * v[0];
}
static void f_unit(const float *v, float *o) { /* ray_tracing.rs:230-237 */
    float len = sqrtf(f_dot(v, v));
    o[0] = v[0] / len; o[1] = v[1] / len; o[2] = v[2] / len;
}

void or32_camera_initialize(RtCamera *c) {
    c->image_height = (uint32_t)((float)c->image_width / c->aspect_ratio); /* ray_tracing.rs:301-302 */
    if (c->image_height < 1) c->image_height = 1;
    c->pixel_samples_scale = 1.0f / (float)c->samples_per_pixel;
    for (int k = 0; k < 3; k++) c->center[k] = c->lookfrom[k];
    float diff[3] = {c->lookfrom[0] - c->lookat[0], c->lookfrom[1] - c->lookat[1], c->lookfrom[2] - c->lookat[2]};
    float focal_length = sqrtf(f_dot(diff, diff));
    const float rads_per_deg = 3.14159265358979323846264338327950288f / 180.0f; /* f32::to_radians */
    float theta = c->vfov * rads_per_deg;
    float h = tanf(theta / 2.0f);
    float viewport_height = 2.0f * h * focal_length;
    float viewport_width = viewport_height * ((float)c->image_width / (float)c->image_height);
    f_unit(diff, c->w);
    float cr[3];
    f_cross(c->vup, c->w, cr);
    f_unit(cr, c->u);
    f_cross(c->w, c->u, c->v);
    float vu[3], vv[3];
    for (int k = 0; k < 3; k++) {
        vu[k] = c->u[k] * viewport_width;
        vv[k] = c->v[k] * -viewport_height;
        c->pixel_delta_u[k] = vu[k] * (1.0f / (float)c->image_width);
        c->pixel_delta_v[k] = vv[k] * (1.0f / (float)c->image_height);
    }
    for (int k = 0; k < 3; k++) {
        float ul = ((c->center[k] - c->w[k] * focal_length) - vu[k] * 0.5f) - vv[k] * 0.5f; /* ray_tracing.rs:324-330 */
        c->pixel00_loc[k] = ul + (c->pixel_delta_u[k] + c->pixel_delta_v[k]) * 0.5f;          /* ray_tracing.rs:332-335 */
    }
}

void or32_camera_from_rt(const RtCamera *c, OrCameraF *o) {
    o->image_width = c->image_width; o->image_height = c->image_height;
    for (int k = 0; k < 3; k++) {
        o->center[k] = c->center[k]; o->pixel00_loc[k] = c->pixel00_loc[k];
        o->pixel_delta_u[k] = c->pixel_delta_u[k]; o->pixel_delta_v[k] = c->pixel_delta_v[k];
    }
}

void or64_camera_from_rt(const RtCamera *c, OrCameraD *o) {
    o->image_width = c->image_width; o->image_height = c->image_height;
    for (int k = 0; k < 3; k++) {
        o->center[k] = c->center[k]; o->pixel00_loc[k] = c->pixel00_loc[k];
        o->pixel_delta_u[k] = c->pixel_delta_u[k]; o->pixel_delta_v[k] = c->pixel_delta_v[k];
    }
}

void or32_camera_from_d(const OrCameraD *c, OrCameraF *o) {
    o->image_width = c->image_width; o->image_height = c->image_height;
    for (int k = 0; k < 3; k++) {
        o->center[k] = (float)c->center[k]; o->pixel00_loc[k] = (float)c->pixel00_loc[k];
        o->pixel_delta_u[k] = (float)c->pixel_delta_u[k]; o->pixel_delta_v[k] = (float)c->pixel_delta_v[k];
    }
}

/* ------------------------------------------------------------------ the path tracer, twice */

#define OR_F32 0
#include "rt_oracle_impl.h"
#undef OR_F32

#define OR_F32 1
#include "rt_oracle_impl.h"
#undef OR_F32

/* ------------------------------------------------------------------ RNG debug views */

static uint8_t to_pixel_u8(float v) { /* ray.cu:81-89 */
    float x = v * 255.0f;
    if (!(x == x)) return 255;
    x = x < 255.0f ? x : 255.0f;
    x = x > 0.0f ? x : 0.0f;
    return (uint8_t)x;
}

void or32_show_random(uint32_t width, uint32_t height, uint64_t seed, int norm, uint8_t *rgba) {
    for (uint32_t p = 0; p < width * height; p++) {
        uint64_t st, inc;
        or_stream_init(&st, &inc, seed, p, 0, 0);
        float rgb[3];
        if (!norm) { /* random_float3, library.cu:18-24 */
            for (int k = 0; k < 3; k++) rgb[k] = or_u01(or_pcg32_next(&st, inc));
        } else {     /* random_norm_float3, library.cu:25-31, with the rejection sampler used by the renderer */
            rgb[0] = 1.0f; rgb[1] = 0.0f; rgb[2] = 0.0f;
            for (int i = 0; i < 20; i++) { /* one try = one output cut into 11 + 11 + 10 bits, as in or32_random_unit_vector */
                uint32_t a = or_pcg32_next(&st, inc);
                float x = fmaf((float)(a >> 21), 0x1p-10f, -1.0f + 0x1p-11f);
                float y = fmaf((float)((a >> 10) & 0x7ffu), 0x1p-10f, -1.0f + 0x1p-11f);
                float z = fmaf((float)(a & 0x3ffu), 0x1p-9f, -1.0f + 0x1p-10f);
                float d = fmaf(z, z, fmaf(y, y, x * x));
                if (d <= 1.0f) {
                    float inv = 1.0f / sqrtf(d);
                    rgb[0] = x * inv; rgb[1] = y * inv; rgb[2] = z * inv;
                    break;
                }
            }
        }
        rgba[4 * p + 0] = to_pixel_u8(rgb[0]);
        rgba[4 * p + 1] = to_pixel_u8(rgb[1]);
        rgba[4 * p + 2] = to_pixel_u8(rgb[2]);
        rgba[4 * p + 3] = to_pixel_u8(1.0f);
    }
}
