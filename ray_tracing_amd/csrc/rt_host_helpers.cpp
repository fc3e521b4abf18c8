// rt_host_helpers.cpp -- the parts of the C-ABI that need no GPU: camera set-up, scene builders,
// PPM output.  Citations are relative to the upstream repository root.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/rt_hip.h"

namespace {

struct Vec3f { float v[3]; };

// f32 vector helpers with the arithmetic of gpu-implementation/src/ray_tracing.rs:218-249
// (plain multiplies and adds; this file is built with -ffp-contract=off).
inline float dot3(const float* u, const float* v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; }
inline void cross3(const float* u, const float* v, float* o) {
    o[0] = u[1] * v[2] - u[2] * v[1];
    o[1] = u[2] * v[0] - u[0] * v[2];
    o[2] = u[0] * v[1] - u[1] * v[0];
}
inline void unit3(const float* v, float* o) {
    float len = std::sqrt(dot3(v, v));
    o[0] = v[0] / len; o[1] = v[1] / len; o[2] = v[2] / len;
}

// Host-side PCG32 for the scene builders (scenes are inputs; their generator only has to be
// deterministic).
struct SceneRng {
    uint64_t state, inc;
    explicit SceneRng(uint64_t seed) : state(0), inc((0xda3e39cb94b95bdbULL << 1) | 1u) {
        next(); state += seed; next();
    }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((0u - rot) & 31u));
    }
    float uniform() { return (float)(next() >> 8) * 0x1p-24f; }                 // [0,1)
    float range(float lo, float hi) { return lo + (hi - lo) * uniform(); }
};

RtSphere make_sphere(float x, float y, float z, float r, uint32_t mat, float ar, float ag, float ab, float param) {
    RtSphere s;
    s.center[0] = x; s.center[1] = y; s.center[2] = z;
    s.radius = r;
    s.material = mat;
    s.albedo[0] = ar; s.albedo[1] = ag; s.albedo[2] = ab;
    if (mat == RT_MAT_REFLECT) param = param < 1.0f ? param : 1.0f;    // Reflect ctor clamp, library.cu:144
    s.param = param;
    return s;
}

uint32_t emit(const std::vector<RtSphere>& v, RtSphere* out, uint32_t cap) {
    uint32_t n = (uint32_t)v.size();
    if (out) std::memcpy(out, v.data(), sizeof(RtSphere) * (n < cap ? n : cap));
    return n;
}

}  // namespace

extern "C" {

int rt_camera_initialize(RtCamera* c) {
    if (!c || c->image_width == 0 || !(c->aspect_ratio > 0.0f)) return RT_ERR_INVALID;
    // gpu-implementation/src/ray_tracing.rs:300-336, statement by statement
    c->image_height = (uint32_t)((float)c->image_width / c->aspect_ratio);
    if (c->image_height < 1) c->image_height = 1;
    c->pixel_samples_scale = 1.0f / (float)c->samples_per_pixel;
    for (int k = 0; k < 3; k++) c->center[k] = c->lookfrom[k];
    float diff[3] = {c->lookfrom[0] - c->lookat[0], c->lookfrom[1] - c->lookat[1], c->lookfrom[2] - c->lookat[2]};
    float focal_length = std::sqrt(dot3(diff, diff));
    const float rads_per_deg = 3.14159265358979323846264338327950288f / 180.0f;   // f32::to_radians
    float theta = c->vfov * rads_per_deg;
    float h = std::tan(theta / 2.0f);
    float viewport_height = 2.0f * h * focal_length;
    float viewport_width = viewport_height * ((float)c->image_width / (float)c->image_height);
    unit3(diff, c->w);
    float cr[3];
    cross3(c->vup, c->w, cr);
    unit3(cr, c->u);
    cross3(c->w, c->u, c->v);
    float vu[3], vv[3];
    for (int k = 0; k < 3; k++) {
        vu[k] = c->u[k] * viewport_width;
        vv[k] = c->v[k] * -viewport_height;
        c->pixel_delta_u[k] = vu[k] * (1.0f / (float)c->image_width);
        c->pixel_delta_v[k] = vv[k] * (1.0f / (float)c->image_height);
    }
    for (int k = 0; k < 3; k++) {
        float upper_left = ((c->center[k] - c->w[k] * focal_length) - vu[k] * 0.5f) - vv[k] * 0.5f;
        c->pixel00_loc[k] = upper_left + (c->pixel_delta_u[k] + c->pixel_delta_v[k]) * 0.5f;
    }
    return 0;
}

int rt_camera_new(float aspect_ratio, uint32_t image_width, RtCamera* out) {
    if (!out) return RT_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    out->aspect_ratio = aspect_ratio;                         // ray_tracing.rs:276-294
    out->image_width = image_width;
    out->samples_per_pixel = 10;
    out->max_depth = 10;
    out->vfov = 90.0f;
    out->lookfrom[0] = -2.0f; out->lookfrom[1] = 2.0f; out->lookfrom[2] = 1.0f;
    out->lookat[0] = 0.0f; out->lookat[1] = 0.0f; out->lookat[2] = -1.0f;
    out->vup[0] = 0.0f; out->vup[1] = 1.0f; out->vup[2] = 0.0f;
    return rt_camera_initialize(out);
}

int rt_camera_new_cpu(double aspect_ratio, uint32_t image_width, RtCamera* out) {
    if (!out || image_width == 0 || !(aspect_ratio > 0.0)) return RT_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    // cpu-implementation/src/lib.rs:148-183 in f64, then rounded to the f32 block
    uint32_t height = (uint32_t)((double)image_width / aspect_ratio);
    if (height == 0) height = 1;
    const double f_width = (double)image_width, f_height = (double)height;
    const double vp_height = 2.0, vp_width = vp_height * aspect_ratio, focal_length = 1.0;
    const double vp[3] = {vp_width, -vp_height, 0.0};
    const double denom[3] = {f_width, f_height, 1.0};
    const double unit_z[3] = {0.0, 0.0, 1.0};
    double delta[3];
    for (int k = 0; k < 3; k++) delta[k] = vp[k] / denom[k];
    out->aspect_ratio = (float)aspect_ratio;
    out->image_width = image_width;
    out->image_height = height;
    out->samples_per_pixel = 1;
    out->pixel_samples_scale = 1.0f;
    out->vfov = 90.0f;
    out->lookat[2] = -1.0f;
    out->vup[1] = 1.0f;
    out->u[0] = 1.0f; out->v[1] = 1.0f; out->w[2] = 1.0f;
    for (int k = 0; k < 3; k++) {
        double top_left = 0.0 - unit_z[k] * focal_length - vp[k] / 2.0;
        out->center[k] = 0.0f;
        out->pixel00_loc[k] = (float)(top_left + delta[k] / 2.0);
    }
    out->pixel_delta_u[0] = (float)delta[0];   // (x, y, 0) * pixel_delta element-wise: lib.rs:215-216
    out->pixel_delta_v[1] = (float)delta[1];
    return 0;
}

uint32_t rt_scene_cpu_default(RtSphere* out, uint32_t cap) {
    std::vector<RtSphere> v;                                  // cpu-implementation/src/main.rs:26-28
    v.push_back(make_sphere(0.0f, 0.0f, -1.0f, 0.5f, RT_MAT_DIFFUSE, 0.5f, 0.5f, 0.5f, 0.0f));
    v.push_back(make_sphere(0.0f, -100.5f, -1.0f, 100.0f, RT_MAT_DIFFUSE, 0.5f, 0.5f, 0.5f, 0.0f));
    return emit(v, out, cap);
}

uint32_t rt_scene_gpu_default(RtSphere* out, uint32_t cap) {
    std::vector<RtSphere> v;                                  // gpu-implementation/src/cuda/ray.cu:97-106
    v.push_back(make_sphere(0.0f, -100.5f, -1.0f, 100.0f, RT_MAT_DIFFUSE, 0.8f, 0.8f, 0.0f, 0.0f));
    v.push_back(make_sphere(0.0f, 0.0f, -1.2f, 0.5f, RT_MAT_DIFFUSE, 0.1f, 0.2f, 0.5f, 0.0f));
    v.push_back(make_sphere(-1.0f, 0.0f, -1.2f, 0.5f, RT_MAT_REFLECT, 0.8f, 0.8f, 0.8f, 0.3f));
    v.push_back(make_sphere(1.0f, 0.0f, -1.2f, 0.5f, RT_MAT_REFLECT, 0.8f, 0.6f, 0.2f, 1.0f));
    v.push_back(make_sphere(1.0f, 0.0f, -3.2f, 0.5f, RT_MAT_REFLECT, 0.8f, 0.6f, 0.2f, 0.0f));
    v.push_back(make_sphere(0.0f, 0.0f, -50.0f, 25.0f, RT_MAT_REFLECT, 0.9f, 0.9f, 0.9f, 0.0f));
    v.push_back(make_sphere(0.0f, 0.0f, -110.0f, 25.0f, RT_MAT_REFLECT, 0.9f, 0.9f, 0.9f, 0.0f));
    v.push_back(make_sphere(2.0f, 0.0f, -80.0f, 0.5f, RT_MAT_DIFFUSE, 0.8f, 0.6f, 0.2f, 0.0f));
    v.push_back(make_sphere(-2.0f, 0.0f, -80.0f, 0.5f, RT_MAT_REFLECT, 0.8f, 0.6f, 0.2f, 0.3f));
    return emit(v, out, cap);
}

uint32_t rt_scene_book1_final(uint64_t seed, RtSphere* out, uint32_t cap) {
    // "Ray Tracing in One Weekend", final scene; draw order: material choice, centre x, centre z, material draws.
    SceneRng g(seed);
    std::vector<RtSphere> v;
    v.push_back(make_sphere(0.0f, -1000.0f, 0.0f, 1000.0f, RT_MAT_DIFFUSE, 0.5f, 0.5f, 0.5f, 0.0f));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            float choose = g.uniform();
            float cx = (float)a + 0.9f * g.uniform();
            float cz = (float)b + 0.9f * g.uniform();
            float dx = cx - 4.0f, dz = cz;
            if (!(std::sqrt(dx * dx + dz * dz) > 0.9f)) continue;
            if (choose < 0.8f) {
                float r = g.uniform() * g.uniform(), gg = g.uniform() * g.uniform(), bb = g.uniform() * g.uniform();
                v.push_back(make_sphere(cx, 0.2f, cz, 0.2f, RT_MAT_DIFFUSE, r, gg, bb, 0.0f));
            } else if (choose < 0.95f) {
                float r = g.range(0.5f, 1.0f), gg = g.range(0.5f, 1.0f), bb = g.range(0.5f, 1.0f);
                float fuzz = g.range(0.0f, 0.5f);
                v.push_back(make_sphere(cx, 0.2f, cz, 0.2f, RT_MAT_REFLECT, r, gg, bb, fuzz));
            } else {
                v.push_back(make_sphere(cx, 0.2f, cz, 0.2f, RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f));
            }
        }
    }
    v.push_back(make_sphere(0.0f, 1.0f, 0.0f, 1.0f, RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f));
    v.push_back(make_sphere(-4.0f, 1.0f, 0.0f, 1.0f, RT_MAT_DIFFUSE, 0.4f, 0.2f, 0.1f, 0.0f));
    v.push_back(make_sphere(4.0f, 1.0f, 0.0f, 1.0f, RT_MAT_REFLECT, 0.7f, 0.6f, 0.5f, 0.0f));
    return emit(v, out, cap);
}

uint32_t rt_scene_grid(uint64_t seed, uint32_t side, RtSphere* out, uint32_t cap) {
    SceneRng g(seed);
    std::vector<RtSphere> v;
    v.reserve((size_t)side * side + 1);
    v.push_back(make_sphere(0.0f, -1000.0f, 0.0f, 1000.0f, RT_MAT_DIFFUSE, 0.5f, 0.5f, 0.5f, 0.0f));
    const float half = 0.5f * (float)side;
    for (uint32_t i = 0; i < side; i++) {
        for (uint32_t j = 0; j < side; j++) {
            float r = g.range(0.2f, 0.45f);
            float cx = (float)i - half + 0.5f + g.range(-0.05f, 0.05f);
            float cz = (float)j - half + 0.5f + g.range(-0.05f, 0.05f);
            float choose = g.uniform();
            float c0 = g.uniform(), c1 = g.uniform(), c2 = g.uniform(), f = g.uniform();
            if (choose < 0.70f)
                v.push_back(make_sphere(cx, r, cz, r, RT_MAT_DIFFUSE, c0, c1, c2, 0.0f));
            else if (choose < 0.95f)
                v.push_back(make_sphere(cx, r, cz, r, RT_MAT_REFLECT, 0.5f + 0.5f * c0, 0.5f + 0.5f * c1, 0.5f + 0.5f * c2, 0.5f * f));
            else
                v.push_back(make_sphere(cx, r, cz, r, RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f));
        }
    }
    return emit(v, out, cap);
}

uint32_t rt_scene_mirror_hall(uint64_t seed, uint32_t n_small, RtSphere* out, uint32_t cap) {
    SceneRng g(seed);
    std::vector<RtSphere> v;
    // the two facing mirrors of the reference's "reflection test" (ray.cu:103-104)
    v.push_back(make_sphere(0.0f, 0.0f, -50.0f, 25.0f, RT_MAT_REFLECT, 0.9f, 0.9f, 0.9f, 0.0f));
    v.push_back(make_sphere(0.0f, 0.0f, -110.0f, 25.0f, RT_MAT_REFLECT, 0.9f, 0.9f, 0.9f, 0.0f));
    for (uint32_t i = 0; i < n_small; i++) {
        float cx = g.range(-6.0f, 6.0f), cy = g.range(-4.0f, 4.0f), cz = g.range(-83.0f, -77.0f);
        float r = g.range(0.3f, 0.7f);
        float choose = g.uniform();
        float c0 = g.uniform(), c1 = g.uniform(), c2 = g.uniform(), f = g.uniform();
        if (choose < 0.5f)
            v.push_back(make_sphere(cx, cy, cz, r, RT_MAT_REFLECT, 0.6f + 0.4f * c0, 0.6f + 0.4f * c1, 0.6f + 0.4f * c2, 0.2f * f));
        else
            v.push_back(make_sphere(cx, cy, cz, r, RT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f));
    }
    return emit(v, out, cap);
}

int rt_write_ppm(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba || !width || !height) return RT_ERR_INVALID;
    FILE* f = std::fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::vector<uint8_t> row((size_t)width * 3);
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t* p = rgba + (size_t)y * width * 4;
        for (uint32_t x = 0; x < width; x++) {
            row[3 * x + 0] = p[4 * x + 0]; row[3 * x + 1] = p[4 * x + 1]; row[3 * x + 2] = p[4 * x + 2];
        }
        if (std::fwrite(row.data(), 1, row.size(), f) != row.size()) { std::fclose(f); return RT_ERR_IO; }
    }
    return std::fclose(f) == 0 ? 0 : RT_ERR_IO;
}

int rt_write_ppm_p3f(const char* path, const float* linear, uint32_t width, uint32_t height) {
    if (!path || !linear || !width || !height) return RT_ERR_INVALID;
    FILE* f = std::fopen(path, "w");
    if (!f) return RT_ERR_IO;
    std::fprintf(f, "P3\n%u %u\n255\n", width, height);
    // Rust's `{}` prints an f64 with the shortest digits that round-trip and never with an exponent
    // (2.55999e-05 is written 0.0000255999), "NaN" for a NaN and "inf" / "-inf": fixed notation here too.
    // The largest double takes 309 digits.
    constexpr size_t kMax = 330;
    char buf[3 * kMax + 4];
    for (size_t k = 0; k < (size_t)width * height; k++) {
        char* q = buf;
        for (int c = 0; c < 3; c++) {
            double v = 255.999 * (double)linear[4 * k + c];
            if (v != v) { std::memcpy(q, "NaN", 3); q += 3; }
            else q = std::to_chars(q, q + kMax, v, std::chars_format::fixed).ptr;
            *q++ = c < 2 ? ' ' : '\n';
        }
        if (std::fwrite(buf, 1, (size_t)(q - buf), f) != (size_t)(q - buf)) { std::fclose(f); return RT_ERR_IO; }
    }
    return std::fclose(f) == 0 ? 0 : RT_ERR_IO;
}

}  // extern "C"
