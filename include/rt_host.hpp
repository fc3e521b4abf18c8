// rt_host.hpp -- header-only C++ mirror of the reference's host interface, over the C-ABI of rt_hip.h.
//
// The reference's host is Rust; this image has no Rust toolchain, so the compiled-language host is C++ with
// the reference's names and argument meaning:
//   rt::Camera::create(aspect, width)        camera::Camera::new            gpu-implementation/src/ray_tracing.rs:275
//   rt::Camera::initialize()                 camera::Camera::initialize     ray_tracing.rs:300
//   rt::Camera::create_cpu(aspect, width)    Camera::new                    cpu-implementation/src/lib.rs:148
//   rt::Sphere{center, radius, material}     Sphere::new / Sphere(c, r, m)  lib.rs:91, library.cu:193
//   rt::Diffuse / rt::Reflect                Diffuse(c) / Reflect(c, fuzz)  library.cu:130,144   (+ rt::Dielectric)
//   rt::DeviceGUI                            DeviceGUI                      ray_tracing.rs:43-63
//   rt::HipWorld{new, compile, render}       CudaWorld::{new, compile_ptx, render}  ray_tracing.rs:105-211
// Errors surface as rt::Error (the reference pushes strings to its GUI or panics).
#ifndef RT_HOST_HPP
#define RT_HOST_HPP

#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_hip.h"

namespace rt {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error("[" + std::to_string(c) + "] " + m), code(c) {}
};

using Vec3 = std::array<float, 3>;

struct Material { uint32_t tag; Vec3 albedo; float param; };
inline Material Diffuse(Vec3 color) { return {RT_MAT_DIFFUSE, color, 0.0f}; }
inline Material Reflect(Vec3 color, float fuzz) { return {RT_MAT_REFLECT, color, fuzz < 1.0f ? fuzz : 1.0f}; }  // library.cu:144
inline Material Dielectric(float ior, Vec3 color = {1.0f, 1.0f, 1.0f}) { return {RT_MAT_DIELECTRIC, color, ior}; }

inline RtSphere Sphere(Vec3 center, float radius, Material m = Diffuse({0.5f, 0.5f, 0.5f})) {
    RtSphere s;
    for (int k = 0; k < 3; k++) { s.center[k] = center[k]; s.albedo[k] = m.albedo[k]; }
    s.radius = radius; s.material = m.tag; s.param = m.param;
    return s;
}
using World = std::vector<RtSphere>;

template <class F, class... A>
inline World build_scene(F fn, A... args) {
    World w(fn(args..., nullptr, 0u));
    fn(args..., w.data(), (uint32_t)w.size());
    return w;
}
inline World scene_cpu_default() { return build_scene(rt_scene_cpu_default); }
inline World scene_gpu_default() { return build_scene(rt_scene_gpu_default); }
inline World scene_book1_final(uint64_t seed = 1) { return build_scene(rt_scene_book1_final, seed); }
inline World scene_grid(uint64_t seed = 2, uint32_t side = 128) { return build_scene(rt_scene_grid, seed, side); }
inline World scene_mirror_hall(uint64_t seed = 3, uint32_t n_small = 64) { return build_scene(rt_scene_mirror_hall, seed, n_small); }

struct Camera {
    RtCamera block{};
    static Camera create(float aspect_ratio, uint32_t image_width) {
        Camera c;
        if (int rc = rt_camera_new(aspect_ratio, image_width, &c.block)) throw Error(rc, "rt_camera_new: bad arguments");
        return c;
    }
    static Camera create_cpu(double aspect_ratio, uint32_t image_width) {
        Camera c;
        if (int rc = rt_camera_new_cpu(aspect_ratio, image_width, &c.block)) throw Error(rc, "rt_camera_new_cpu: bad arguments");
        return c;
    }
    static Camera look_at(float aspect_ratio, uint32_t image_width, float vfov, Vec3 from, Vec3 at, Vec3 up = {0.0f, 1.0f, 0.0f}) {
        Camera c = create(aspect_ratio, image_width);
        c.block.vfov = vfov;
        for (int k = 0; k < 3; k++) { c.block.lookfrom[k] = from[k]; c.block.lookat[k] = at[k]; c.block.vup[k] = up[k]; }
        c.initialize();
        return c;
    }
    void initialize() {
        if (int rc = rt_camera_initialize(&block)) throw Error(rc, "rt_camera_initialize: bad camera");
    }
    uint32_t image_width() const { return block.image_width; }
    uint32_t image_height() const { return block.image_height; }
};

// DeviceGUI::default() of ray_tracing.rs:53-63 (4x4 grid samples, depth 4) plus the GPU path's t range (ray.cu:24).
inline RtParams DeviceGUI(uint32_t sample2_per_pixel = 4, uint32_t max_depth = 4, uint32_t block_dim = 0) {
    RtParams p{};
    p.samples_per_pixel = sample2_per_pixel * sample2_per_pixel;
    p.sample2_per_pixel = sample2_per_pixel;
    p.max_depth = max_depth;
    p.flags = RT_FLAG_SAMPLER_GRID;
    p.t_min = 0.001f;
    p.t_max = 10024.0f;
    p.block_dim = block_dim;
    return p;
}
// Jittered samples (lib.rs:208-219) with the GPU path's shading.
inline RtParams JitterParams(uint32_t samples_per_pixel, uint32_t max_depth, uint64_t seed = 0, uint32_t sample_chunk = 0) {
    RtParams p = DeviceGUI(1, max_depth);
    p.samples_per_pixel = samples_per_pixel;
    p.flags = 0;
    p.seed = seed;
    p.sample_chunk = sample_chunk;
    return p;
}
// cpu-implementation behaviour (lib.rs:184-242; main.rs:16-19 defaults 64 spp, depth 20).
inline RtParams CpuParams(uint32_t samples_per_pixel = 64, uint32_t max_depth = 20, uint64_t seed = 0) {
    RtParams p = JitterParams(samples_per_pixel, max_depth, seed);
    p.flags = RT_FLAG_BAND_ALBEDO | RT_FLAG_FLIP_SIGNBIT | RT_FLAG_QUANT_ROUND;
    p.t_max = INFINITY;
    return p;
}

class HipWorld {
public:
    explicit HipWorld(int device = 0) {
        if (int rc = rt_create(device, &ctx_)) throw Error(rc, rt_last_error(nullptr));
    }
    HipWorld(int device, const World& w) : HipWorld(device) { upload_scene(w); }
    ~HipWorld() { rt_destroy(ctx_); }
    HipWorld(const HipWorld&) = delete;
    HipWorld& operator=(const HipWorld&) = delete;

    void compile(const std::vector<uint8_t>* image = nullptr) {        // compile_ptx: (re)load a code object
        check(image ? rt_load_module(ctx_, image->data(), image->size()) : rt_load_module(ctx_, nullptr, 0));
    }
    void upload_scene(const World& w) { check(rt_upload_scene(ctx_, w.data(), (uint32_t)w.size())); }

    // Camera::render(frame, world, max_depth) / CudaWorld::render(frame, camera): fills `frame` (RGBA8, W*H*4 bytes)
    uint64_t render(uint8_t* frame, const Camera& cam, const RtParams& p, float* linear = nullptr, int32_t* hits = nullptr) {
        uint64_t rays = 0;
        check(rt_render_rows(ctx_, &cam.block, &p, 0, cam.image_height(), frame, linear, hits, &rays));
        return rays;
    }
    std::vector<uint8_t> render(const Camera& cam, const RtParams& p, uint64_t* rays = nullptr) {
        std::vector<uint8_t> frame((size_t)cam.image_width() * cam.image_height() * 4);
        uint64_t r = render(frame.data(), cam, p);
        if (rays) *rays = r;
        return frame;
    }
    std::vector<uint8_t> show_random(uint32_t w, uint32_t h, uint64_t seed, bool norm) {
        std::vector<uint8_t> frame((size_t)w * h * 4);
        check(rt_show_random(ctx_, w, h, seed, norm ? 1 : 0, frame.data()));
        return frame;
    }
    // One frame over several GPUs (one HipWorld per GPU and rank): rank 0 makes an id, every rank joins, then
    // all call render_sharded; rank 0's frame holds the result.
    static std::vector<uint8_t> comm_id() {
        std::vector<uint8_t> id(RT_COMM_ID_BYTES);
        if (int rc = rt_comm_id(id.data())) throw Error(rc, rt_last_error(nullptr));
        return id;
    }
    void comm_init(const std::vector<uint8_t>& id, uint32_t world, uint32_t rank) {
        if (id.size() != RT_COMM_ID_BYTES) throw Error(RT_ERR_INVALID, "communicator id must be RT_COMM_ID_BYTES long");
        check(rt_comm_init(ctx_, id.data(), world, rank));
    }
    uint64_t render_sharded(uint8_t* frame, const Camera& cam, const RtParams& p, uint32_t tile_rows = 8) {
        uint64_t rays = 0;
        check(rt_render_frame_sharded(ctx_, &cam.block, &p, tile_rows, frame, &rays));
        return rays;
    }
    float last_kernel_ms() const { return rt_last_kernel_ms(ctx_); }
    rt_ctx* raw() { return ctx_; }

private:
    void check(int rc) { if (rc) throw Error(rc, rt_last_error(ctx_)); }
    rt_ctx* ctx_ = nullptr;
};

inline void write_ppm(const std::string& path, const std::vector<uint8_t>& rgba, uint32_t w, uint32_t h) {
    if (int rc = rt_write_ppm(path.c_str(), rgba.data(), w, h)) throw Error(rc, "rt_write_ppm(" + path + ") failed");
}

}  // namespace rt
#endif
