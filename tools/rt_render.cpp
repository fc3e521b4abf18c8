// rt_render -- command-line front end: renders one of the built-in scenes on one or more GPUs and writes a PPM.
// It is the headless counterpart of the reference's two window apps (cpu-implementation/src/main.rs,
// gpu-implementation/src/main.rs): same scenes, cameras and defaults, no window.
//   rt_render --scene cpu|gpu|book1|grid|hall [--width W] [--spp N | --grid n] [--depth D] [--seed S]
//             [--chunk C] [--out image.ppm] [--p3f linear.ppm] [--gpus N [--devices d0,d1,...]] [--frames K]
// --gpus N (> 1): one host thread and one context per rank; the frame is dealt out as interleaved 8-row tiles and
// gathered to rank 0 with RCCL (rt_render_frame_sharded).  Rank r runs on GPU r, or on the r-th entry of --devices;
// several ranks can share a GPU only with a transport that allows it (RCCL does not; the stand-in the tests load
// through RT_RCCL_LIB does).  --frames K renders the frame K times and reports the last.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

#include "../include/rt_host.hpp"

static const char* arg(int argc, char** argv, const char* name, const char* def) {
    for (int i = 1; i + 1 < argc; i++) if (!std::strcmp(argv[i], name)) return argv[i + 1];
    return def;
}

int main(int argc, char** argv) {
    try {
        std::string scene = arg(argc, argv, "--scene", "cpu");
        uint32_t width = (uint32_t)std::atoi(arg(argc, argv, "--width", scene == "cpu" ? "256" : "1280"));
        uint32_t depth = (uint32_t)std::atoi(arg(argc, argv, "--depth", scene == "cpu" ? "20" : "4"));
        uint32_t spp = (uint32_t)std::atoi(arg(argc, argv, "--spp", scene == "cpu" ? "64" : "0"));
        uint32_t grid = (uint32_t)std::atoi(arg(argc, argv, "--grid", "4"));
        uint64_t seed = std::strtoull(arg(argc, argv, "--seed", "0"), nullptr, 10);
        uint32_t chunk = (uint32_t)std::atoi(arg(argc, argv, "--chunk", "0"));
        std::string out = arg(argc, argv, "--out", "image.ppm");
        std::string p3f = arg(argc, argv, "--p3f", "");
        uint32_t gpus = (uint32_t)std::atoi(arg(argc, argv, "--gpus", "1"));
        uint32_t frames = (uint32_t)std::atoi(arg(argc, argv, "--frames", "1"));
        if (gpus == 0 || frames == 0) { std::fprintf(stderr, "--gpus and --frames must be >= 1\n"); return 2; }

        rt::World world;
        rt::Camera cam;
        RtParams p;
        if (scene == "cpu") {                                      // cpu-implementation/src/main.rs:16-34
            world = rt::scene_cpu_default();
            cam = rt::Camera::create_cpu(16.0 / 9.0, width);
            p = rt::CpuParams(spp ? spp : 64, depth, seed);
        } else {
            if (scene == "gpu") { world = rt::scene_gpu_default(); cam = rt::Camera::create(16.0f / 9.0f, width); }   // gpu main.rs:22-25
            else if (scene == "book1") { world = rt::scene_book1_final(1); cam = rt::Camera::look_at(16.0f / 9.0f, width, 20.0f, {13, 2, 3}, {0, 0, 0}); }
            else if (scene == "grid") { world = rt::scene_grid(2, 128); cam = rt::Camera::look_at(16.0f / 9.0f, width, 40.0f, {0, 30, 80}, {0, 0, 0}); }
            else if (scene == "hall") { world = rt::scene_mirror_hall(3, 64); cam = rt::Camera::look_at(16.0f / 9.0f, width, 90.0f, {0, 0, -85}, {0, 0, -70}); }
            else { std::fprintf(stderr, "unknown scene %s\n", scene.c_str()); return 2; }
            p = spp ? rt::JitterParams(spp, depth, seed) : rt::DeviceGUI(grid, depth);
            p.seed = seed;
        }
        p.sample_chunk = chunk;

        std::vector<uint8_t> frame((size_t)cam.image_width() * cam.image_height() * 4);
        if (gpus > 1) {
            if (!p3f.empty()) { std::fprintf(stderr, "--p3f needs --gpus 1\n"); return 2; }
            int have = 0;
            std::vector<int> device(gpus);
            for (uint32_t r = 0; r < gpus; r++) device[r] = (int)r;
            const std::string list = arg(argc, argv, "--devices", "");
            if (!list.empty()) {
                size_t at = 0;
                for (uint32_t r = 0; r < gpus; r++) {
                    if (at >= list.size()) { std::fprintf(stderr, "--devices needs %u entries\n", gpus); return 2; }
                    device[r] = std::atoi(list.c_str() + at);
                    at = list.find(',', at);
                    at = at == std::string::npos ? list.size() : at + 1;
                }
            }
            if (rt_device_count(&have) != 0) { std::fprintf(stderr, "rt_render: %s\n", rt_last_error(nullptr)); return 2; }
            for (uint32_t r = 0; r < gpus; r++)
                if (device[r] < 0 || device[r] >= have) {                    // a missing rank would leave the others waiting in RCCL
                    std::fprintf(stderr, "rt_render: rank %u wants GPU %d but %d GPU(s) are visible\n", r, device[r], have);
                    return 2;
                }
            const std::vector<uint8_t> id = rt::HipWorld::comm_id();
            std::vector<uint64_t> rays(gpus, 0);
            std::vector<float> kernel_ms(gpus, 0.0f);
            std::vector<std::string> errors(gpus);
            std::vector<double> wall(gpus, 0.0);
            // every context (GPU set-up, scene upload) exists before any rank enters the collective comm_init: a GPU
            // that cannot be set up ends the program here instead of leaving the other ranks waiting for it
            std::vector<std::unique_ptr<rt::HipWorld>> ctx;
            for (uint32_t r = 0; r < gpus; r++) ctx.push_back(std::make_unique<rt::HipWorld>(device[r], world));
            std::vector<std::thread> ranks;
            for (uint32_t r = 0; r < gpus; r++)
                ranks.emplace_back([&, r] {
                    try {
                        rt::HipWorld& hip = *ctx[r];
                        hip.comm_init(id, gpus, r);
                        for (uint32_t f = 0; f < frames; f++) {
                            auto t0 = std::chrono::steady_clock::now();
                            rays[r] = hip.render_sharded(r == 0 ? frame.data() : nullptr, cam, p);
                            wall[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                        }
                        kernel_ms[r] = hip.last_kernel_ms();
                    } catch (const std::exception& e) { errors[r] = e.what(); }
                });
            for (auto& t : ranks) t.join();
            uint64_t total = 0;
            float slowest = 0.0f;
            for (uint32_t r = 0; r < gpus; r++) {
                if (!errors[r].empty()) throw std::runtime_error("GPU " + std::to_string(r) + ": " + errors[r]);
                total += rays[r];
                if (kernel_ms[r] > slowest) slowest = kernel_ms[r];
            }
            std::printf("Calculation took %.3f ms on %u GPUs (slowest kernel %.3f ms): %ux%u, %zu spheres, %u spp, depth %u, %llu ray segments, %.1f Mrays/s\n",
                        wall[0] * 1e3, gpus, slowest, cam.image_width(), cam.image_height(), world.size(), p.samples_per_pixel,
                        p.max_depth, (unsigned long long)total, total / (wall[0] * 1e6));
            rt::write_ppm(out, frame, cam.image_width(), cam.image_height());
            return 0;
        }
        rt::HipWorld hip(0, world);
        std::vector<float> linear(p3f.empty() ? 0 : frame.size());
        uint64_t rays = 0;
        double wall = 0.0;
        for (uint32_t f = 0; f < frames; f++) {
            auto t0 = std::chrono::steady_clock::now();
            rays = hip.render(frame.data(), cam, p, linear.empty() ? nullptr : linear.data());
            wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        std::printf("Calculation took %.3f ms (kernel %.3f ms): %ux%u, %zu spheres, %u spp, depth %u, %llu ray segments, %.1f Mrays/s\n",
                    wall * 1e3, hip.last_kernel_ms(), cam.image_width(), cam.image_height(), world.size(), p.samples_per_pixel,
                    p.max_depth, (unsigned long long)rays, rays / (hip.last_kernel_ms() * 1e3));   // cpu main.rs:62-64 prints the time
        rt::write_ppm(out, frame, cam.image_width(), cam.image_height());
        if (!p3f.empty()) {
            if (int rc = rt_write_ppm_p3f(p3f.c_str(), linear.data(), cam.image_width(), cam.image_height())) throw rt::Error(rc, "rt_write_ppm_p3f failed");
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rt_render: %s\n", e.what());
        return 1;
    }
}
