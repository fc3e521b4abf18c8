// rt_hip.cpp -- device half of the C-ABI declared in include/rt_hip.h: a thin layer over
// hipModuleLoadData / hipModuleLaunchKernel / hipMalloc / hipMemcpy driving the gfx950 code object
// built from rt_kernels.hip (embedded below with .incbin).  It takes the place of the reference's
// cudarc-based CudaWorld (gpu-implementation/src/ray_tracing.rs:73-211).  Host C++ only.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>   // types only: the RCCL entry points are resolved at run time (dlopen), see Rccl below
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_device.h"
#include "rt_scratch.h"

static_assert(RT_COUNTER_BYTES <= RT_SCRATCH_BYTES, "the kernels' counter block must fit the static scratch array");

// The code object produced by `hipcc --offload-arch=gfx950 --genco rt_kernels.hip`.
__asm__(
    ".section .rodata\n"
    ".balign 4096\n"
    ".global rt_embedded_hsaco\n"
    "rt_embedded_hsaco:\n"
    ".incbin \"rt_kernels.hsaco\"\n"
    ".global rt_embedded_hsaco_end\n"
    "rt_embedded_hsaco_end:\n"
    ".byte 0\n"
    ".previous\n");
extern "C" const unsigned char rt_embedded_hsaco[];
extern "C" const unsigned char rt_embedded_hsaco_end[];

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipModule_t module = nullptr;
    hipFunction_t f_render = nullptr, f_render_queue = nullptr, f_render_queue_fold = nullptr, f_render_queue_stream_fold = nullptr, f_render_stream = nullptr, f_render_queue_stream = nullptr, f_hits = nullptr, f_random = nullptr, f_assemble = nullptr, f_camera_terms = nullptr, f_render_pool = nullptr, f_render_pool_fold = nullptr, f_render_pool_plain = nullptr, f_render_queue_boxes = nullptr, f_render_pool_boxes = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *d_geom = nullptr, *d_exact = nullptr, *d_shade = nullptr, *d_mat = nullptr;
    void* d_camterms = nullptr;         // per sphere {A, B, C, sq} of the camera-ray scan, rewritten by every launch (rt_camera_terms)
    float translate[3] = {0.0f, 0.0f, 0.0f};
    // Scale of the scan filter (rt_kernels.hip, scan_tile): sigma = 2^-k with sigma * 4 R <= 1, R = the largest
    // |centre - translate| + radius of the scene; scene_radius = R, or a negative number when the scene has no usable
    // bound (non-finite or absurd values): such scenes stay off the kernels whose bounce scan relies on the bound.
    float scan_scale = 1.0f;
    double scene_radius = 0.0;
    uint32_t n_spheres = 0, n_padded = 0;
    uint32_t n_blocks = 0;            // blocks of 32 device spheres with a bounding box behind the filter data; 0: the list is not sorted
    bool have_scene = false;
    void *d_rgba = nullptr, *d_linear = nullptr, *d_hits = nullptr, *d_counter = nullptr;
    size_t counter_need = RT_COUNTER_BYTES, cap_counter = 0;   // counter block the loaded code object asks for; bytes d_counter holds
    void* d_scratch = nullptr;          // this GPU's copy of the library's static device array (not owned; rt_scratch.hip)
    int scratch_rc = 0;                 // why d_scratch is NULL, if it is
    // scheduling experiments (rt_debug_schedule): forced unit shape (0 = the library's choice), item kernel for sample_chunk 1
    uint32_t dbg_unit_w = 0, dbg_unit_h = 0;
    bool dbg_no_queue = false;
    int dbg_scheduler = 0;                                       // RT_SCHED_*: 0 = the library's choice between the two
    size_t cap_rgba = 0, cap_linear = 0, cap_hits = 0;
    void* d_fold = nullptr;             // fold-order workspace of the queue kernels: finished samples until their unit is complete
    size_t cap_fold = 0;
    size_t fold_limit = (size_t)8 << 30; // largest fold-order workspace the context will allocate (rt_debug_fold_workspace_limit)
    bool cull_lists = true, cull_boxes = true;   // rt_debug_cull
    // multi-GPU gather: this context is rank `rank` of `world`
    ncclComm_t comm = nullptr;
    uint32_t world = 1, rank = 0;
    void* d_status = nullptr;                        // three 32-bit words for the ranks' status exchange (rt_comm_init):
                                                     // [0] this rank's status, [1] the all-reduce's result, [2] constant 1
    void *d_tiles = nullptr, *d_frame = nullptr;     // rank's tile buffer (rank 0: all ranks' buffers), assembled frame
    size_t cap_tiles = 0, cap_frame = 0;
    float last_ms = 0.0f;
    const char* last_kernel = "";
    void* last_counters = nullptr;      // counter block of the last render launch
    int compute_units = 0, clock_khz = 0;
    char name[600] = {0};
    std::string err;
};

// message of the last failure that had no context to carry it (rt_create, rt_comm_id, ...), per calling thread
static thread_local std::string g_create_error;

namespace {

int fail(rt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

int hip_fail(rt_ctx* c, hipError_t e, const char* what) {
    // The runtime also keeps the error as this thread's "last error"; the code is reported to the caller here, so
    // it is cleared there: other users of the runtime in this process (PyTorch checks hipGetLastError after its
    // own launches) must not find it later and take it for theirs.
    (void)hipGetLastError();
    std::string m = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
    return fail(c, (int)e, m);
}

#define RT_HIP(ctx, call)                                         \
    do {                                                          \
        hipError_t e__ = (call);                                  \
        if (e__ != hipSuccess) return hip_fail((ctx), e__, #call); \
    } while (0)

int ensure(rt_ctx* c, void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return 0;
    if (*p) { RT_HIP(c, hipFree(*p)); *p = nullptr; *cap = 0; }
    RT_HIP(c, hipMalloc(p, bytes));
    *cap = bytes;
    return 0;
}

int load_image(rt_ctx* c, const void* image) {
    hipModule_t m = nullptr;
    RT_HIP(c, hipModuleLoadData(&m, image));
    hipFunction_t fr, fq, fs, fqs, fh, fx, fa, fqf, fqsf, fct;
    hipError_t e;
    if ((e = hipModuleGetFunction(&fr, m, "rt_render")) != hipSuccess ||
        (e = hipModuleGetFunction(&fq, m, "rt_render_queue")) != hipSuccess ||
        (e = hipModuleGetFunction(&fqs, m, "rt_render_queue_stream")) != hipSuccess ||
        (e = hipModuleGetFunction(&fqf, m, "rt_render_queue_fold")) != hipSuccess ||
        (e = hipModuleGetFunction(&fqsf, m, "rt_render_queue_stream_fold")) != hipSuccess ||
        (e = hipModuleGetFunction(&fs, m, "rt_render_stream")) != hipSuccess ||
        (e = hipModuleGetFunction(&fh, m, "rt_primary_hits")) != hipSuccess ||
        (e = hipModuleGetFunction(&fx, m, "rt_show_random")) != hipSuccess ||
        (e = hipModuleGetFunction(&fct, m, "rt_camera_terms")) != hipSuccess ||
        (e = hipModuleGetFunction(&fa, m, "rt_assemble")) != hipSuccess) {
        (void)hipModuleUnload(m);
        return hip_fail(c, e, "hipModuleGetFunction");
    }
    // the counter block this object's kernels expect (objects without the word: the layout of rt_device.h)
    size_t need = RT_COUNTER_BYTES;
    {
        hipDeviceptr_t sym = nullptr;
        size_t sym_bytes = 0;
        uint32_t word = 0;
        if (hipModuleGetGlobal(&sym, &sym_bytes, m, "rt_counter_bytes") == hipSuccess && sym && sym_bytes == sizeof word) {
            if ((e = hipMemcpy(&word, sym, sizeof word, hipMemcpyDeviceToHost)) != hipSuccess) {
                (void)hipModuleUnload(m);
                return hip_fail(c, e, "hipMemcpy(rt_counter_bytes)");
            }
            if (word < RT_COUNTER_BYTES || word > (1u << 20) || word % 8u != 0u) {
                (void)hipModuleUnload(m);
                return fail(c, RT_ERR_INVALID, "code object states an implausible counter block size");
            }
            need = word;
        } else {
            (void)hipGetLastError();
        }
    }
    // the tile size of the object's streaming kernels: rt_upload_scene lays the scene out for RT_Q_STREAM_TILE, and those
    // kernels find the arrays from it (objects without the word: built before the layout depended on it -- refused too)
    {
        hipDeviceptr_t sym = nullptr;
        size_t sym_bytes = 0;
        uint32_t word = 0;
        if (hipModuleGetGlobal(&sym, &sym_bytes, m, "rt_stream_tile") != hipSuccess || !sym || sym_bytes != sizeof word) {
            (void)hipGetLastError();
            (void)hipModuleUnload(m);
            return fail(c, RT_ERR_INVALID, "code object does not state the tile size of its streaming kernels (rt_stream_tile)");
        }
        if ((e = hipMemcpy(&word, sym, sizeof word, hipMemcpyDeviceToHost)) != hipSuccess) {
            (void)hipModuleUnload(m);
            return hip_fail(c, e, "hipMemcpy(rt_stream_tile)");
        }
        if (word != RT_Q_STREAM_TILE) {
            (void)hipModuleUnload(m);
            return fail(c, RT_ERR_INVALID, "code object was built for " + std::to_string(word) + "-sphere tiles, this library for " + std::to_string(RT_Q_STREAM_TILE));
        }
    }
    if (need > c->cap_counter) {
        void* fresh = nullptr;
        if ((e = hipMalloc(&fresh, need)) != hipSuccess) { (void)hipModuleUnload(m); return hip_fail(c, e, "hipMalloc(counter block)"); }
        if (c->d_counter) (void)hipFree(c->d_counter);
        if (c->last_counters == c->d_counter) c->last_counters = nullptr;
        c->d_counter = fresh;
        c->cap_counter = need;
    }
    c->counter_need = need;
    if (c->module) (void)hipModuleUnload(c->module);
    c->module = m;
    if (hipModuleGetFunction(&c->f_render_pool, m, "rt_render_pool") != hipSuccess) { c->f_render_pool = nullptr; (void)hipGetLastError(); }
    if (hipModuleGetFunction(&c->f_render_pool_fold, m, "rt_render_pool_fold") != hipSuccess) { c->f_render_pool_fold = nullptr; (void)hipGetLastError(); }
    if (hipModuleGetFunction(&c->f_render_pool_plain, m, "rt_render_pool_plain") != hipSuccess) { c->f_render_pool_plain = nullptr; (void)hipGetLastError(); }
    if (hipModuleGetFunction(&c->f_render_queue_boxes, m, "rt_render_queue_boxes") != hipSuccess) { c->f_render_queue_boxes = nullptr; (void)hipGetLastError(); }
    if (hipModuleGetFunction(&c->f_render_pool_boxes, m, "rt_render_pool_boxes") != hipSuccess) { c->f_render_pool_boxes = nullptr; (void)hipGetLastError(); }
    c->f_render_queue_fold = fqf; c->f_render_queue_stream_fold = fqsf; c->f_render = fr; c->f_render_queue = fq; c->f_render_queue_stream = fqs; c->f_render_stream = fs; c->f_hits = fh; c->f_random = fx; c->f_assemble = fa; c->f_camera_terms = fct;
    return 0;
}

int check_common(rt_ctx* c, const RtCamera* cam, const RtParams* p) {
    if (!c) return RT_ERR_INVALID;
    if (!cam || !p) return fail(c, RT_ERR_INVALID, "camera or params is NULL");
    if (!c->module) return fail(c, RT_ERR_NO_MODULE, "no code object loaded");
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    if (cam->image_width == 0 || cam->image_height == 0) return fail(c, RT_ERR_INVALID, "camera not initialised (zero image size)");
    if ((uint64_t)cam->image_width * cam->image_height >= (1ull << 31)) return fail(c, RT_ERR_INVALID, "image has 2^31 pixels or more");
    if (!(p->t_min >= 0.0f)) return fail(c, RT_ERR_INVALID, "t_min must be >= 0");
    if (!(p->t_max > p->t_min)) return fail(c, RT_ERR_INVALID, "t_max must be > t_min");
    if (p->block_dim != 0 && (p->block_dim % 64 != 0 || p->block_dim < 64 || p->block_dim > 1024))
        return fail(c, RT_ERR_INVALID, "block_dim must be a multiple of 64 in 64..1024");
    return 0;
}

int check_render(rt_ctx* c, const RtParams* p) {
    if (p->samples_per_pixel == 0) return fail(c, RT_ERR_INVALID, "samples_per_pixel must be >= 1");
    if (p->flags & RT_FLAG_SAMPLER_GRID) {
        if (p->sample2_per_pixel == 0 || (uint64_t)p->sample2_per_pixel * p->sample2_per_pixel != p->samples_per_pixel)
            return fail(c, RT_ERR_INVALID, "grid sampler needs samples_per_pixel == sample2_per_pixel^2");
    }
    return 0;
}

// can this launch run on the camera-ray-batch kernels?  (every sample its own work item)
bool queue_kernel_fits(const rt_ctx* c, const RtCamera* cam, uint32_t spp) {
    if (c->dbg_no_queue || spp < 2u || (uint64_t)64u * spp >= (1ull << 31) ||
        c->n_padded > 65536u)                                    // a queued ray names the sphere it left in 16 bits
        return false;
    // Their bounce scan assumes that a bounce ray starts within R (1 + small) of the scene's translation point (it
    // starts on a sphere).  A hit point carries the rounding of o + t d, which grows with the camera's distance: a
    // camera farther than 2^12 R away (or a scene without a bound) keeps the launch on the other kernels.
    if (c->n_spheres == 0u) return true;
    if (!(c->scene_radius > 0.0)) return false;
    double dist2 = 0.0;
    for (int k = 0; k < 3; k++) {
        const double dk = (double)cam->center[k] - (double)c->translate[k];
        dist2 += dk * dk;
    }
    return dist2 <= 4096.0 * 4096.0 * c->scene_radius * c->scene_radius;   // false for NaN
}

// fold_queue: sample_chunk = 0 (the reference's fp32 fold) may run on the camera-ray-batch kernels with a workspace
// for the finished samples; false keeps it on the one-lane-per-pixel kernel, which needs none.
RtLaunch make_launch(const rt_ctx* c, const RtCamera* cam, const RtParams* p, uint32_t row_base,
                     uint32_t rows_local, uint32_t tile_rows, uint32_t tile_stride_rows, bool fold_queue = false) {
    RtLaunch L;
    std::memset(&L, 0, sizeof L);
    L.n_spheres = c->n_spheres;
    L.n_padded = c->n_padded;
    L.samples_per_pixel = p->samples_per_pixel;
    L.sample2_per_pixel = p->sample2_per_pixel ? p->sample2_per_pixel : 1u;
    L.max_depth = p->max_depth;
    L.flags = p->flags;
    L.seed_lo = (uint32_t)(p->seed & 0xffffffffu);
    L.seed_hi = (uint32_t)(p->seed >> 32);
    L.t_min = p->t_min;
    L.t_max = p->t_max;
    L.width = cam->image_width;
    L.height = cam->image_height;
    L.row_base = row_base;
    L.rows_local = rows_local;
    L.tile_rows = tile_rows;
    L.tile_stride_rows = tile_stride_rows;
    L.tiles_x = (cam->image_width + 7u) / 8u;
    L.n_wave_tiles = L.tiles_x * ((rows_local + 7u) / 8u);
    // work decomposition of the render kernels
    const uint32_t spp = p->samples_per_pixel ? p->samples_per_pixel : 1u;
    const bool fold = p->sample_chunk == 0 || p->sample_chunk >= spp;
    if (fold && !(fold_queue && queue_kernel_fits(c, cam, spp))) {
        L.chunk = spp; L.n_chunks = 1; L.unit_w = L.unit_h = 8;         // one lane per pixel, fp32 fold
    } else {
        L.fold = fold ? 1u : 0u;                                        // samples traced in any order, added in sample order
        L.chunk = fold ? 1u : p->sample_chunk;
        L.n_chunks = (spp + L.chunk - 1u) / L.chunk;
        // Unit shape only affects scheduling (the sums are exact).  4x1 rather than 2x2: a unit is written as one
        // 16-byte row segment (2x2: two 8-byte ones, 5x the frame's bytes in HBM writes).  A unit must lie in one row
        // tile: any height does when the rows are contiguous.
        auto height_ok = [&](uint32_t h) { return tile_stride_rows == tile_rows || tile_rows % h == 0u; };
        L.unit_w = 4u; L.unit_h = 1u;
        // A wavefront holds two units at a time and cannot open a third before the older one's last, longest paths have
        // ended, so a unit must be enough work to outlast them: measured on the camera-ray-batch kernels, units of 2,000 to
        // 4,000 samples run at full speed, 512 samples 20 % below it (book scene, 128 spp, 4x1 against 16x1 units).  Units
        // grow (wider first: one row segment per unit is written) until they hold 2,048 samples -- but only while the
        // launch still has 80 units per resident wavefront (24 per compute unit): the end of the launch, when nothing is
        // left to take, lasts about two units (4K book frame at 500 spp: the whole frame runs 1 % faster with 8x1 units,
        // 169 per wavefront, a quarter of it 3 % slower with 42 per wavefront than with 4x1 and 84).
        // Below 512 samples a unit grows regardless: it has to fill its camera-ray batches.
        const uint64_t resident = (uint64_t)(c->compute_units > 0 ? c->compute_units : 256) * 24u;
        auto n_units_with = [&](uint32_t w, uint32_t h) {
            return (uint64_t)((cam->image_width + w - 1u) / w) * ((rows_local + h - 1u) / h);
        };
        const bool compact = c->n_padded > RT_TILE_SPHERES;
        auto grow = [&](uint64_t min_units) {
            L.unit_w = 4u; L.unit_h = 1u;
            while (L.unit_w * L.unit_h < 64u) {
                const uint64_t items = (uint64_t)L.unit_w * L.unit_h * L.n_chunks, samples = (uint64_t)L.unit_w * L.unit_h * spp;
                uint32_t w = L.unit_w, h = L.unit_h;
                // Streamed lists: 8 wide, then two rows, then 16 wide -- the bounce rays of a compact unit start close
                // together and reach fewer tiles and blocks (16,385-sphere frame 122 -> 115 ms with 8x2 against 8x1 units).
                // Otherwise 16 wide first: a unit's row is then a whole 64-byte line (8x2 units measured 1 % faster on the
                // book 4K frame and 0.4 % on the 8K mirror hall, but there the half lines of neighbouring units, written
                // at different times from different XCDs, showed as 170 MB more HBM writes per frame).
                if (w < 8u) w *= 2u;
                else if (compact && h < 2u && height_ok(2u)) h = 2u;
                else if (w < 16u || !height_ok(2u * h)) w *= 2u;
                else h *= 2u;
                // (a unit is written as one row segment per row and taken with one atomic add on the work counter, which
                // the memory side executes: the HBM write counter shows the frame once plus 32 bytes per unit, at any unit
                // shape.  A unit that is large enough still widens to 16 pixels, a whole 64-byte line per store and half
                // the atomics of 8; measured equal in time at every launch size.)
                if (items >= 512u && ((samples >= 2048u && L.unit_w * L.unit_h >= 16u) || n_units_with(w, h) < min_units)) break;
                L.unit_w = w; L.unit_h = h;
            }
        };
        // Short launches run with the units pooled per workgroup (rt_render_pool, see launch_render): the end of the launch,
        // when the wavefronts run out of units one by one, then costs little -- whole workgroups finish the last units --
        // and the units can have their full size.  In steady state the pooled kernel is 7-8 % behind (its bookkeeping per
        // camera-ray batch weighs more since the units' sphere lists shortened the batches, and it holds more
        // wave-uniform state than there are scalar registers).  The crossover is a launch length, roughly 70-100 ms of
        // kernel time, estimated here as samples per resident wavefront (24 per compute unit) x (spheres + 56) < 44 M.
        // Measured with 16x1 units (kernel ms, per-wavefront / pooled; one box): book scene 4K at 500 spp, depth 50 (486
        // spheres, bounce scan with block boxes): half the frame 256 / 277, a quarter (169 k samples per wavefront x 542 =
        // 91 M) 129 / 139, an eighth (46 M) 67.8 / 69.9, a sixteenth (23 M) 41.0 / 37.3; mirror hall 8K at 256 spp, depth
        // 100 (66 spheres): whole frame 349 / 373, half (691 k x 122 = 84 M) 180 / 187, a quarter (42 M) 96.8 / 94.1, an
        // eighth 49.3 / 47.7, a sixteenth 24.9 / 24.2.
        const uint64_t launch_samples = (uint64_t)cam->image_width * rows_local * spp;
        const bool small_launch = launch_samples * (c->n_padded + 56u) < resident * 44000000ull;
        const bool pool_able = L.chunk == 1u && L.n_chunks > 1u && queue_kernel_fits(c, cam, spp) && p->max_depth > 0u &&
                               c->n_padded <= RT_TILE_SPHERES && (fold ? c->f_render_pool_fold : c->f_render_pool) != nullptr &&
                               c->dbg_scheduler != RT_SCHED_WAVE;
        grow(0u);
        if (pool_able && (small_launch || c->dbg_scheduler == RT_SCHED_POOL)) L.pool_slots = 1u;
        // (a streamed list: a round through the tiles costs the same whatever the units, and 8x2 units at 25 per wavefront
        // beat 8x1 at 50 -- 16,385-sphere frame at 64 spp 122 -> 115 ms)
        else grow((c->n_padded > RT_TILE_SPHERES ? 20u : 80u) * resident);
        // Sphere lists streamed through LDS: a pass over the whole list takes the better part of a millisecond, a unit
        // of 512 samples some 25 of them, and the wavefronts of a launch finish a unit apart.  With fewer than 10 such
        // units per resident wavefront (20 per compute unit on that kernel) the units are halved once: measured on the
        // 16,385-sphere frame at 64 spp, an eighth of the frame (6 units of 8x1 per wavefront) 179 -> 172 ms and a
        // sixteenth 107 -> 97 ms with 4x1 units, a quarter (13 per wavefront) 323 -> 326 ms, half 617 -> 629 ms.
        if (c->n_padded > RT_TILE_SPHERES && L.pool_slots == 0u && L.chunk == 1u && L.n_chunks > 1u && L.unit_w >= 8u &&
            (uint64_t)L.unit_w * L.unit_h * spp >= 512u &&
            n_units_with(L.unit_w, L.unit_h) < (uint64_t)10u * (uint64_t)(c->compute_units > 0 ? c->compute_units : 256) * 20u)
            L.unit_w /= 2u;
        if (c->dbg_unit_w && height_ok(c->dbg_unit_h)) {                                                        // rt_debug_schedule
            L.unit_w = c->dbg_unit_w; L.unit_h = c->dbg_unit_h;
            if (pool_able && c->dbg_scheduler == RT_SCHED_AUTO) L.pool_slots = small_launch ? 1u : 0u;
        }
        // fold-order workspace: 2 units x P pixels x spp colours of 16 bytes per wavefront, at most 32 wavefronts per
        // compute unit resident; beyond the context's limit (8 GiB unless changed) the fold stays on the
        // one-lane-per-pixel kernel, which needs no workspace and adds the same samples in the same order
        const uint64_t fold_bytes = (uint64_t)(c->compute_units > 0 ? c->compute_units : 256) * 32u * 2u * L.unit_w * L.unit_h * spp * 16u;
        if (L.fold && fold_bytes > (uint64_t)c->fold_limit) {
            L.fold = 0u; L.chunk = spp; L.n_chunks = 1; L.unit_w = L.unit_h = 8; L.pool_slots = 0u;
        }
    }
    for (int k = 0; k < 3; k++) L.translate[k] = c->translate[k];
    L.scan_scale = c->scan_scale;
    L.sorted = c->n_blocks != 0u ? 1u : 0u;
    L.scan_blocks = c->cull_boxes ? c->n_blocks : 0u;
    if (std::getenv("RT_EXP_NO_BOXES")) L.scan_blocks = 0u;                      // timing experiments only
    L.units_bottom_up = 0u;
    if (const char* e = std::getenv("RT_EXP_UNITS_BOTTOM_UP")) L.units_bottom_up = std::atoi(e) ? 1u : 0u;   // timing experiments only
    for (L.unit_shift = 0; (1u << L.unit_shift) < L.unit_w; L.unit_shift++) {}
    for (L.unit_pshift = 0; (1u << L.unit_pshift) < L.unit_w * L.unit_h; L.unit_pshift++) {}
    L.units_x = (cam->image_width + L.unit_w - 1u) / L.unit_w;
    L.n_units = L.units_x * ((rows_local + L.unit_h - 1u) / L.unit_h);
    // Sphere lists per pixel unit for the camera-ray batches (unit_list in rt_kernels.hip): 16 entries where the
    // workgroup's LDS has the room (launch_render).  unit_reach: the largest distance between the direction of a camera
    // ray of a unit, as make_ray rounds it, and the exact direction through the centre of the unit's footprint:
    // (w + 1/2)/2 |du| + (h + 1/2)/2 |dv| + eta, eta bounding make_ray's four roundings per component (each below 2^-24
    // of a partial result: 2^-24 (|pixel00_k| + |pixel00_k - o_k| + 3 (W+1) |du_k| + 4 (H+1) |dv_k|), as in
    // rt_camera_terms), rounded up.
    L.ulist_cap = 0u;
    L.unit_reach = 0.0f;
    // (a list costs some 400 operations per unit to set up and saves 4 per sphere and batch: not worth it for a handful of spheres)
    if (L.n_chunks > 1u && L.chunk == 1u && c->n_padded >= 16u && queue_kernel_fits(c, cam, spp)) {
        double nu = 0.0, nv = 0.0, eta = 0.0;
        const double W1 = (double)cam->image_width + 1.0, H1 = (double)cam->image_height + 1.0;
        for (int k = 0; k < 3; k++) {
            const double du = cam->pixel_delta_u[k], dv = cam->pixel_delta_v[k];
            nu += du * du;
            nv += dv * dv;
            const double e = 0x1p-24 * (std::fabs((double)cam->pixel00_loc[k]) + std::fabs((double)cam->pixel00_loc[k] - (double)cam->center[k]) +
                                        3.0 * W1 * std::fabs(du) + 4.0 * H1 * std::fabs(dv));
            eta += e * e;
        }
        const double reach = (0.5 * (L.unit_w + 0.5) * std::sqrt(nu) + 0.5 * (L.unit_h + 0.5) * std::sqrt(nv) + std::sqrt(eta)) * (1.0 + 0x1p-20);
        if (reach == reach && reach < 1e30) {
            L.unit_reach = std::nextafter((float)reach, INFINITY);
            L.ulist_cap = 16u;
        }
    }
    if (!c->cull_lists) L.ulist_cap = 0u;
    if (const char* e = std::getenv("RT_EXP_ULIST_CAP")) {                       // timing experiments and tests only: 0 .. 16
        const unsigned v = (unsigned)std::atoi(e);
        if (v < L.ulist_cap) L.ulist_cap = v & ~1u;
    }
    return L;
}

void launch_geometry(const RtParams* p, uint32_t n_wave_units, unsigned* grid, unsigned* block) {
    *block = p->block_dim ? p->block_dim : 256u;
    unsigned waves = *block / 64u;
    *grid = (n_wave_units + waves - 1u) / waves;
}

// Dynamic LDS per workgroup: the sphere tile (filter float4 + exact float4 per sphere) plus, per wavefront, one
// candidate-list block and the pixel accumulators of its two open units.
size_t lds_bytes(const rt_ctx* c, unsigned block, unsigned unit_pixels) {
    uint32_t n = c->n_padded < RT_TILE_SPHERES ? c->n_padded : RT_TILE_SPHERES;
    return (size_t)n * 32u + (size_t)(block / 64u) * (RT_CAND_CAP * 64u * 2u + 2u * 3u * unit_pixels * 8u);
}

// One render launch into device buffers; blocking; fills last_ms and (optionally) the segment count.
// `counters`: RT_COUNTER_BYTES of device memory for the launch-wide counters (the context's own block, or the
// library's static array).  Allocates nothing.
int launch_render(rt_ctx* c, const RtCamera* cam, const RtParams* p, RtLaunch L, void* d_rgba, void* d_linear,
                  uint64_t* rays_out, void* counters, bool fold_workspace = true) {
    unsigned grid, block;
    launch_geometry(p, L.n_units, &grid, &block);
    unsigned waves = block / 64u;
    const unsigned P = L.unit_w * L.unit_h;
    hipFunction_t f;
    unsigned lds;
    if ((uint64_t)P * L.n_chunks >= (1ull << 31)) return fail(c, RT_ERR_INVALID, "too many work items per pixel unit (samples_per_pixel / sample_chunk too large)");
    const bool queue = L.n_chunks > 1 && L.chunk == 1 && queue_kernel_fits(c, cam, L.samples_per_pixel);
    if (L.fold && !queue) return fail(c, RT_ERR_INVALID, "internal: fold-order launch prepared for a kernel that cannot run it");
    if (queue) {
        // every sample its own item: camera rays traced in batches, bounce rays from a queue
        const bool one_tile = c->n_padded <= RT_TILE_SPHERES;
        if (L.fold) {
            f = one_tile ? c->f_render_queue_fold : c->f_render_queue_stream_fold;
            c->last_kernel = one_tile ? "rt_render_queue_fold" : "rt_render_queue_stream_fold";
        } else {
            f = one_tile ? c->f_render_queue : c->f_render_queue_stream;
            c->last_kernel = one_tile ? "rt_render_queue" : "rt_render_queue_stream";
            if (one_tile && L.scan_blocks != 0u && c->f_render_queue_boxes) {   // the build with the blocks' boxes in the bounce scan
                f = c->f_render_queue_boxes;
                c->last_kernel = "rt_render_queue_boxes";
            } else if (one_tile && L.pool_slots == 0u) {
                L.scan_blocks = 0u;
            }
        }
        if (one_tile && p->block_dim == 0u) {
            // the one-tile kernels are built for six waves per SIMD; with 512-thread workgroups three of them (24
            // wavefronts) fit a compute unit's LDS on the 485-sphere scene, with 256 threads only five (20 wavefronts)
            block = 512u;
            waves = block / 64u;
            grid = (L.n_units + waves - 1u) / waves;
        }
        if (!one_tile && p->block_dim == 0u) {
            // the streaming kernel's wavefronts move through the tiles in lock step, and most of a round was waiting for
            // the workgroup's slowest at the barriers: workgroups of ONE wavefront wait for nobody, and with 256-sphere
            // tiles 18 of them share a compute unit's LDS (16,385-sphere frame, 512-sphere tiles: 64 threads 94 ms, 128:
            // 87, 192: 106, 256: 111, 512: 186; 256-sphere tiles: 64 threads 74 ms, 128: 90)
            block = 64u;
            waves = block / 64u;
            grid = (L.n_units + waves - 1u) / waves;
        }
        // the unit lists are the first thing to go where a workgroup's LDS (64 KiB) is short: 16 entries, 12, 8, none
        auto queue_lds = [&](unsigned cap) {
            return (one_tile ? c->n_padded * 32u + L.scan_blocks * 32u : RT_Q_STREAM_TILE * 16u + (L.scan_blocks ? RT_Q_STREAM_TILE + 32u + 16u : 0u)) +
                   waves * (RT_Q_CAND_CAP * 128u + 48u * P + RT_Q_WORDS * 256u + RT_UL_BYTES(cap));
        };
        auto pool_lds = [&](unsigned cap) {
            return c->n_padded * 32u + L.scan_blocks * 32u + 2u * waves * 32u + 16u + 2u * waves * 24u * P + 2u * waves * RT_UL_BYTES(cap) + waves * (RT_Q_CAND_CAP * 128u + RT_Q_WORDS * 256u);
        };
        // Which build.  The per-wavefront kernel and the pooled one each come with and without the blocks' boxes in the
        // bounce scan (fold order: with), the pooled one also without the units' lists: code a launch does not use still
        // costs it registers.
        hipFunction_t pool_f = nullptr;
        const char* pool_name = nullptr;
        if (L.pool_slots != 0u) {
            // the boxes and the lists are the first things to go where a workgroup's LDS (64 KiB) is short
            while (L.ulist_cap != 0u && pool_lds(L.ulist_cap) > 65536u) L.ulist_cap = L.ulist_cap > 12u ? 12u : L.ulist_cap > 8u ? 8u : 0u;
            if (pool_lds(0u) > 65536u) L.scan_blocks = 0u;
            if (L.fold) { pool_f = c->f_render_pool_fold; pool_name = "rt_render_pool_fold"; }
            else if (L.scan_blocks != 0u && c->f_render_pool_boxes) { pool_f = c->f_render_pool_boxes; pool_name = "rt_render_pool_boxes"; }
            else { pool_f = c->f_render_pool; pool_name = "rt_render_pool"; L.scan_blocks = 0u; }
        }
        if (L.ulist_cap != 0u && one_tile) {
            // (and the lists must not cost a workgroup per compute unit: a wave of occupancy is worth more.  Not so where
            // the list is streamed: without a unit list every camera ray scans the whole scene's camera-ray terms in
            // global memory -- measured five times the frame time)
            hipFunction_t fq = L.pool_slots ? pool_f : f;
            auto resident_with = [&](unsigned cap) {
                const unsigned bytes = L.pool_slots ? pool_lds(cap) : queue_lds(cap);
                int n = 0;
                if (bytes > 65536u || hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, fq, (int)block, bytes) != hipSuccess) n = 0;
                // (measured on the streaming kernel: five 256-thread workgroups are resident up to 31,648 bytes each, four
                // from 31,776 on, where the runtime's figure still says five: 155 KiB of the 160 are to be had)
                const int by_lds = (int)(158720u / (bytes ? bytes : 1u));
                return n < by_lds ? n : by_lds;
            };
            const int without = resident_with(0u);
            while (L.ulist_cap != 0u && resident_with(L.ulist_cap) < without) L.ulist_cap = L.ulist_cap > 12u ? 12u : L.ulist_cap > 8u ? 8u : 0u;
        }
        lds = queue_lds(L.ulist_cap);
        // Who holds the open pixel units: each wavefront two of its own (rt_render_queue), or the workgroup a pool of
        // them from which all its wavefronts take camera rays (rt_render_pool).  The pool costs about 22 more
        // instructions per batch and trip (level on the 4K book frame, 3 % slower on the 8K mirror hall), but no wavefront
        // waits for its own units' last paths and the launch's last units are finished by whole workgroups: it wins on
        // small launches (an eighth of the book frame 3 %, a sixteenth 18 %).  Decided with the unit shape in make_launch.
        if (L.pool_slots != 0u) {
            f = pool_f;
            c->last_kernel = pool_name;
            if (!L.fold && L.ulist_cap == 0u && L.scan_blocks == 0u && c->f_render_pool_plain) {   // neither lists nor boxes
                f = c->f_render_pool_plain;
                c->last_kernel = "rt_render_pool_plain";
            }
            // as many open units as the wavefronts would hold between them, so the same LDS: slot headers, the
            // control words, the slots' accumulators, and per wavefront the candidate lists and the ray queue
            L.pool_slots = 2u * waves;
            lds = pool_lds(L.ulist_cap);
            grid = L.n_units;                                    // a unit is work for a whole workgroup
        }
    } else {
        f = (c->n_padded <= RT_TILE_SPHERES) ? c->f_render : c->f_render_stream;
        c->last_kernel = (c->n_padded <= RT_TILE_SPHERES) ? "rt_render" : "rt_render_stream";
        lds = (unsigned)lds_bytes(c, block, P);
    }
    // The wavefronts take units from a counter until none are left, so the grid only has to fill the GPU
    // once: as many workgroups as can be resident, if the frame has more units than that.
    int resident = 0;
    RT_HIP(c, hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&resident, f, (int)block, lds));
    if (resident < 1) return fail(c, RT_ERR_INVALID, "render kernel does not fit a compute unit with this block_dim and scene");
    if (std::getenv("RT_DEBUG_LAUNCH"))                                        // launch shape, for experiments
        std::fprintf(stderr, "rt launch: %s block %u lds %u B resident %d per CU, units %ux%u, lists %u, blocks %u, grid %u\n", c->last_kernel, block, lds, resident, L.unit_w, L.unit_h, L.ulist_cap, L.scan_blocks, grid);
    const unsigned fill = (unsigned)resident * (unsigned)(c->compute_units > 0 ? c->compute_units : 1);
    if (grid > fill) grid = fill;
    // fold-order workspace: per wavefront of the grid, two open units x P pixels x samples_per_pixel colours
    void* d_fold = nullptr;
    if (L.fold) {
        const size_t need = (size_t)grid * waves * 2u * P * L.samples_per_pixel * 16u;
        if (!fold_workspace) return fail(c, RT_ERR_INVALID, "internal: fold-order launch without a workspace");
        if (need > c->fold_limit || ensure(c, &c->d_fold, &c->cap_fold, need) != 0) {
            // No room for the workspace (hundreds of MB to several GB at high sample counts): the fold runs on the
            // one-lane-per-pixel kernel instead, which needs none -- same bits, slower -- rather than failing.
            RtLaunch plain = make_launch(c, cam, p, L.row_base, L.rows_local, L.tile_rows, L.tile_stride_rows, false);
            return launch_render(c, cam, p, plain, d_rgba, d_linear, rays_out, counters, false);
        }
        d_fold = c->d_fold;
    }
    // counter block: [0] path segments (+ diagnostics in [1..7]), 64-bit each; byte 64: the unit counter
    void* d_count = rays_out ? counters : nullptr;
    void* d_work = static_cast<char*>(counters) + 64;
    RT_HIP(c, hipMemsetAsync(counters, 0, c->counter_need, c->stream));
    RtCamera cam_copy = *cam;
    void* args[] = {&c->d_geom, &c->d_exact, &c->d_shade, &c->d_mat, &d_rgba, &d_linear, &d_count, &d_work, &d_fold, &c->d_camterms, &cam_copy, &L};
    RT_HIP(c, hipEventRecord(c->ev0, c->stream));
    if (grid > 0 && queue) {
        // the camera-ray scan's four numbers per sphere for this camera (inside the timed region: part of the frame)
        void* targs[] = {&c->d_exact, &c->d_camterms, &cam_copy, &L};
        RT_HIP(c, hipModuleLaunchKernel(c->f_camera_terms, 1, 1, 1, 1024, 1, 1, 0, c->stream, targs, nullptr));   // one workgroup: it reduces over the scene
    }
    if (grid > 0)
        RT_HIP(c, hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, lds, c->stream, args, nullptr));
    RT_HIP(c, hipEventRecord(c->ev1, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipEventElapsedTime(&c->last_ms, c->ev0, c->ev1));
    if (rays_out) RT_HIP(c, hipMemcpy(rays_out, counters, 8, hipMemcpyDeviceToHost));
    c->last_counters = counters;
    return 0;
}

int launch_hits(rt_ctx* c, const RtCamera* cam, const RtParams* p, RtLaunch L, void* d_hits) {
    unsigned grid, block;
    launch_geometry(p, L.n_wave_tiles, &grid, &block);
    RtCamera cam_copy = *cam;
    void* args[] = {&c->d_geom, &c->d_exact, &c->d_mat, &d_hits, &cam_copy, &L};
    if (grid > 0)
        RT_HIP(c, hipModuleLaunchKernel(c->f_hits, grid, 1, 1, block, 1, 1, (unsigned)lds_bytes(c, block, 64u), c->stream, args, nullptr));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

// RCCL, bound at first use so that single-GPU users never load it and a process that already holds an RCCL
// (PyTorch-ROCm ships its own) keeps using that one.  RT_RCCL_LIB names the library to load.
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

void load_rccl(Rccl& r);

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { load_rccl(r); });
    return &r;
}

void load_rccl(Rccl& r) {
    const char* env = std::getenv("RT_RCCL_LIB");
    const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::vector<const char*> names;
    if (env && *env) names.push_back(env);                 // an explicit name is the only one tried
    else names.assign(std::begin(defaults), std::end(defaults));
    std::string why;
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
        const char* e = dlerror();                         // one call: it returns the message and clears it
        if (why.empty()) why = e ? e : "not found";
    }
    if (!r.handle) { r.error = "cannot load RCCL: " + (why.empty() ? std::string("not found") : why); return; }
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(r.handle, name); if (!p) { ok = false; r.error = std::string("RCCL lacks ") + name; } return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(r.handle); r.handle = nullptr; }
}

int rccl_fail(rt_ctx* c, ncclResult_t e, const char* what) {
    Rccl* r = rccl();
    return fail(c, RT_ERR_COMM, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}

#define RT_RCCL(ctx, call)                                             \
    do {                                                               \
        ncclResult_t e__ = (call);                                     \
        if (e__ != ncclSuccess) return rccl_fail((ctx), e__, #call);   \
    } while (0)

// gathered tile buffers (rank after rank, rows_per_rank rows each) -> frame in image order, on the context's stream
int launch_assemble(rt_ctx* c, void* d_gathered, void* d_frame, uint32_t W, uint32_t H, uint32_t tile_rows,
                    uint32_t world, uint32_t rows_per_rank) {
    void* args[] = {&d_gathered, &d_frame, &W, &H, &tile_rows, &world, &rows_per_rank};
    const size_t px = (size_t)W * H;
    unsigned grid = (unsigned)std::min<size_t>((px + 255u) / 256u, (size_t)(c->compute_units > 0 ? c->compute_units : 256) * 16u);
    RT_HIP(c, hipModuleLaunchKernel(c->f_assemble, grid, 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
    return 0;
}

// tiles of `tile_rows` rows in an image of `height` rows owned by rank r of n: r, r+n, ...
uint32_t tiles_of_rank(uint32_t height, uint32_t tile_rows, uint32_t r, uint32_t n) {
    uint32_t total = (height + tile_rows - 1u) / tile_rows;
    return r < total ? (total - r + n - 1u) / n : 0u;
}

}  // namespace

extern "C" {

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_device_count(int* count) {
    if (!count) return RT_ERR_INVALID;
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceCount");
    return 0;
}

int rt_create(int device, rt_ctx** out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceCount");
    if (device < 0 || device >= count) return fail(nullptr, RT_ERR_INVALID, "no such GPU device");
    rt_ctx* c = new rt_ctx();
    c->device = device;
    int rc = 0;
    auto bail = [&](int code) { g_create_error = c->err; rt_destroy(c); return code; };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(hip_fail(c, e, "hipSetDevice"));
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail(hip_fail(c, e, "hipGetDeviceProperties"));
    std::snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
    c->compute_units = prop.multiProcessorCount;
    c->clock_khz = prop.clockRate;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(hip_fail(c, e, "hipStreamCreate"));
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess)
        return bail(hip_fail(c, e, "hipEventCreate"));
    // this GPU's address of the static scratch array, looked up here, at set-up, with the GPU current; a failure
    // only disables rt_render_tiles_device_static
    c->scratch_rc = rt_scratch_lookup(device, &c->d_scratch, nullptr);
    if ((rc = load_image(c, rt_embedded_hsaco)) != 0) return bail(rc);
    *out = c;
    return 0;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(c->comm);
    void* bufs[] = {c->d_geom /* holds the scene's other arrays too */, c->d_rgba, c->d_linear, c->d_hits, c->d_counter, c->d_tiles, c->d_frame, c->d_status, c->d_fold};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->module) (void)hipModuleUnload(c->module);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int rt_load_module(rt_ctx* c, const void* image, size_t size) {
    if (!c) return RT_ERR_INVALID;
    RT_HIP(c, hipSetDevice(c->device));
    if (!image) return load_image(c, rt_embedded_hsaco);
    if (size < 64) return fail(c, RT_ERR_INVALID, "code object image too small");
    return load_image(c, image);
}

int rt_upload_scene(rt_ctx* c, const RtSphere* spheres, uint32_t n) {
    if (!c) return RT_ERR_INVALID;
    if (n && !spheres) return fail(c, RT_ERR_INVALID, "spheres is NULL");
    if (n > (1u << 24)) return fail(c, RT_ERR_INVALID, "more than 2^24 spheres");
    RT_HIP(c, hipSetDevice(c->device));
    // the scans take 8 spheres per step and finish with a step of 4 and one of 2: lists are padded to an even count
    uint32_t pad = RT_SCAN_PAD;
    if (const char* e = std::getenv("RT_EXP_SCAN_PAD")) { if (std::atoi(e) == 8) pad = 8u; }   // timing experiments only
    uint32_t n_padded = ((n + pad - 1u) / pad) * pad;
    if (n_padded == 0) n_padded = pad;
    // geom:  scan-filter data {c - T, m} per sphere: T = component-wise median of the centres,
    //        m = r*r(1+2^-17) - |c-T|^2 (1-2^-18), rounded up (see scan_tile in rt_kernels.hip)
    // exact: {c, r*r} per sphere for the exact test
    // shade: {cx,cy,cz,radius} {albedo rgb, fuzz|ior} per sphere, then {1/ior, r0 in, r0 out, 0} per sphere (glass only);
    //        mat: material tag per sphere
    float T[3] = {0.0f, 0.0f, 0.0f};
    if (n) {
        std::vector<float> v(n);
        for (int k = 0; k < 3; k++) {
            for (uint32_t i = 0; i < n; i++) v[i] = spheres[i].center[k];
            std::nth_element(v.begin(), v.begin() + n / 2, v.end());
            T[k] = std::isfinite(v[n / 2]) ? v[n / 2] : 0.0f;
        }
    }
    // R = largest |c - T| + |radius|, and the filter's scale sigma = 2^-k <= 1 / (4 R)
    double R = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        double cc = 0.0;
        for (int k = 0; k < 3; k++) {
            const double ck = (double)(float)((double)spheres[i].center[k] - (double)T[k]);
            cc += ck * ck;
        }
        const double reach = std::sqrt(cc) + std::fabs((double)spheres[i].radius);
        if (!(reach <= R)) R = reach;                             // a NaN sticks
    }
    float sigma = 1.0f;
    double scene_radius = -1.0;
    if (n == 0) {
        scene_radius = 0.0;
    } else if (std::isfinite(R) && R >= 0x1p-40 && R <= 0x1p40) {
        int k = 0;
        (void)std::frexp(4.0 * R, &k);                            // 4 R = f * 2^k, f in [0.5, 1): 2^-k * 4 R < 1
        sigma = (float)std::ldexp(1.0, -k);
        scene_radius = R;
    }
    const double s2 = (double)sigma * (double)sigma;
    // Device order: spheres sorted along a space-filling curve, so that 32 consecutive ones -- one block of the scans --
    // lie close together and the block's bounding box is small (scan_tile skips a block no lane's ray can reach).
    // The few spheres that are large against the scene (a ground sphere) come first: their blocks' boxes are large
    // whatever their neighbours.  order[i] = the caller's index of device sphere i.
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    uint32_t n_big = 0;                                              // sorted lists: the large spheres, first in `order`
    // (from one whole block of small spheres on: the 66-sphere mirror hall, two blocks and its two mirrors apart, runs 5 %
    // faster with boxes than with every ray against every sphere; RT_EXP_SORT_MIN: timing experiments only)
    const uint32_t sort_min = std::getenv("RT_EXP_SORT_MIN") ? (uint32_t)std::atoi(std::getenv("RT_EXP_SORT_MIN")) : 32u;
    const bool sorted = n >= sort_min && scene_radius > 0.0 && !std::getenv("RT_EXP_NO_SORT");
    if (sorted) {
        float lo[3], hi[3];
        std::vector<float> v(n);
        double extent = 0.0;
        for (int k = 0; k < 3; k++) {
            for (uint32_t i = 0; i < n; i++) v[i] = spheres[i].center[k];
            std::nth_element(v.begin(), v.begin() + n / 50, v.end());
            lo[k] = v[n / 50];
            std::nth_element(v.begin(), v.begin() + (n - 1 - n / 50), v.end());
            hi[k] = v[n - 1 - n / 50];
            extent = std::max(extent, (double)hi[k] - (double)lo[k]);
        }
        if (!(extent > 0.0) || !std::isfinite(extent)) extent = 1.0;
        auto spread = [](uint32_t x) {                               // 10 bits -> every third bit
            x &= 0x3ffu;
            x = (x | (x << 16)) & 0x030000ffu;
            x = (x | (x << 8)) & 0x0300f00fu;
            x = (x | (x << 4)) & 0x030c30c3u;
            x = (x | (x << 2)) & 0x09249249u;
            return x;
        };
        std::vector<uint64_t> key(n);
        for (uint32_t i = 0; i < n; i++) {
            uint32_t q[3];
            for (int k = 0; k < 3; k++) {
                double t = ((double)spheres[i].center[k] - (double)lo[k]) / extent * 1024.0;
                if (!(t > 0.0)) t = 0.0;
                if (t > 1023.0) t = 1023.0;
                q[k] = (uint32_t)t;
            }
            const bool big = std::fabs((double)spheres[i].radius) > extent / 8.0;
            const uint64_t morton = big ? 0u : (1ull << 30) | spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
            if (big) n_big++;
            key[i] = (morton << 32) | i;
        }
        std::sort(key.begin(), key.end());
        for (uint32_t i = 0; i < n; i++) order[i] = (uint32_t)(key[i] & 0xffffffffu);
        // The small spheres (behind the n_big large ones in `order`) are then put in the order of a median-split tree
        // instead: the set is cut across one axis into two parts of whole blocks of 32 (whole tiles of
        // RT_Q_STREAM_TILE while it is larger than one), and so on down to single blocks.  Blocks of exactly 32
        // consecutive spheres of a Z-order curve straddle the curve's cells and come out ragged; these are boxes cut
        // to measure, and fewer of them are reached by a ray.
        if (!std::getenv("RT_EXP_MORTON")) {
            struct Cut { uint32_t first, count; };
            std::vector<Cut> todo;
            todo.push_back({n_big, n - n_big});
            while (!todo.empty()) {
                const Cut cut = todo.back();
                todo.pop_back();
                if (cut.count <= 32u) continue;
                const uint32_t granule = cut.count > RT_Q_STREAM_TILE ? RT_Q_STREAM_TILE : 32u;
                const uint32_t parts = (cut.count + granule - 1u) / granule;
                const uint32_t left = (parts / 2u) * granule;                      // whole blocks (tiles) on the left
                // the cut: of the three axes the one whose two parts' boxes (spheres with their radii) have the smallest
                // total surface
                auto first = order.begin() + cut.first;
                auto surface = [&](uint32_t from, uint32_t to) {
                    double lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
                    for (uint32_t i = from; i < to; i++)
                        for (int k = 0; k < 3; k++) {
                            const double c = spheres[order[i]].center[k], r = std::fabs((double)spheres[order[i]].radius);
                            lo3[k] = std::min(lo3[k], c - r);
                            hi3[k] = std::max(hi3[k], c + r);
                        }
                    const double e[3] = {hi3[0] - lo3[0], hi3[1] - lo3[1], hi3[2] - lo3[2]};
                    return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
                };
                int axis = 0;
                double best = INFINITY;
                for (int k = 0; k < 3; k++) {
                    std::nth_element(first, first + left, first + cut.count, [&](uint32_t a, uint32_t b) {
                        const float ca = spheres[a].center[k], cb = spheres[b].center[k];
                        return ca < cb || (ca == cb && a < b);
                    });
                    const double area = surface(cut.first, cut.first + left) + surface(cut.first + left, cut.first + cut.count);
                    if (area < best) { best = area; axis = k; }
                }
                std::nth_element(first, first + left, first + cut.count, [&](uint32_t a, uint32_t b) {
                    const float ca = spheres[a].center[axis], cb = spheres[b].center[axis];
                    return ca < cb || (ca == cb && a < b);
                });
                todo.push_back({cut.first, left});
                todo.push_back({cut.first + left, cut.count - left});
            }
        }
    }
    // Device slots.  A large sphere's box is reached by nearly every ray, and a block that holds one is scanned by all 64
    // lanes of nearly every wavefront -- 32 spheres for the sake of one.  So the large spheres get the end of the list to
    // themselves: the small ones in curve order, padding up to a whole block, then the large ones (padded to an even
    // count): the scans' last, short step, which has no box and is taken by every lane.  slot[i] = the caller's index of
    // device sphere i, or -1: padding (never a candidate, never hit).  Not done where it would push a list that fits one
    // LDS tile beyond it, nor for more than 30 large spheres (they would fill a block anyway).
    std::vector<int64_t> slot;
    {
        const uint32_t n_small = n - n_big;
        const uint32_t small_padded = ((n_small + 31u) / 32u) * 32u;
        const uint32_t apart = small_padded + ((n_big + pad - 1u) / pad) * pad;
        const bool big_apart = sorted && n_big >= 1u && n_big <= 30u && n_small >= 32u && !(n_padded <= RT_TILE_SPHERES && apart > RT_TILE_SPHERES) &&
                               !std::getenv("RT_EXP_BIG_FIRST");
        if (big_apart) {
            n_padded = apart;
            slot.assign(n_padded, -1);
            for (uint32_t i = 0; i < n_small; i++) slot[i] = order[n_big + i];
            for (uint32_t i = 0; i < n_big; i++) slot[small_padded + i] = order[i];
        } else {
            slot.assign(n_padded, -1);
            for (uint32_t i = 0; i < n; i++) slot[i] = order[i];
        }
    }
    // after the spheres' filter data: per block of 32 spheres two float4, the corners of a box (coordinates relative to
    // T) that holds every point at which the exact test can report a hit of a sphere of the block, for rays leaving
    // from T: each sphere's own |r| (1 + 2^-10) + 2^-9 |c - T| + 2^-19 R beyond its centre (the kernel adds 2^-9 |o - T|
    // for its ray; the exact test's tolerance is below 2^-11 |c - o|)
    const uint32_t n_blocks = (n_padded + 31u) / 32u;
    const uint32_t n_tiles = (n_padded + RT_Q_STREAM_TILE - 1u) / RT_Q_STREAM_TILE;   // ... and one box per tile of the streaming kernels behind those
    std::vector<float> geom(((size_t)n_padded + 2u * n_blocks + 2u * n_tiles + n_blocks) * 4), exact((size_t)n_padded * 4), shade((size_t)n_padded * 12, 0.0f);
    std::vector<uint32_t> mat((size_t)n_padded * 2u);
    for (uint32_t b = 0; b < n_blocks + n_tiles; b++) {
        float* lo4 = &geom[4 * ((size_t)n_padded + 2u * b)];
        lo4[0] = lo4[1] = lo4[2] = INFINITY; lo4[3] = 0.0f;
        lo4[4] = lo4[5] = lo4[6] = -INFINITY; lo4[7] = 0.0f;
    }
    for (uint32_t i = 0; i < n_padded; i++) {
        float* g = &geom[4 * (size_t)i];
        float* x = &exact[4 * (size_t)i];
        float* s = &shade[8 * (size_t)i];
        mat[(size_t)n_padded + i] = slot[i] >= 0 ? (uint32_t)slot[i] : 0u;
        if (slot[i] >= 0) {
            const RtSphere& sp = spheres[slot[i]];
            if (sp.material < RT_MAT_DIFFUSE || sp.material > RT_MAT_DIELECTRIC)
                return fail(c, RT_ERR_INVALID, "unknown material tag in sphere " + std::to_string(slot[i]));
            const float rr = sp.radius * sp.radius;   // radius*radius, library.cu:203 / lib.rs:101
            double cc = 0.0;
            for (int k = 0; k < 3; k++) {
                g[k] = (float)((double)sp.center[k] - (double)T[k]);
                cc += (double)g[k] * (double)g[k];
                x[k] = sp.center[k];
            }
            const double m = (double)rr * (1.0 + 0x1p-17) - cc * (1.0 - 0x1p-18);
            g[3] = std::nextafter((float)m, INFINITY) * (float)s2;   // times sigma^2, a power of two: exact
            {
                float* lo4 = &geom[4 * ((size_t)n_padded + 2u * (i / 32u))];
                const double ext = std::fabs((double)sp.radius) * (1.0 + 0x1p-10) + 0x1p-9 * std::sqrt(cc) + 0x1p-19 * (scene_radius > 0.0 ? scene_radius : 0.0);
                for (int k = 0; k < 3; k++) {
                    const float a = std::nextafter((float)((double)g[k] - ext), -INFINITY), b = std::nextafter((float)((double)g[k] + ext), INFINITY);
                    if (!(a >= lo4[k])) lo4[k] = a;                   // (a NaN sticks: the box then never excludes anything)
                    if (!(b <= lo4[4 + k])) lo4[4 + k] = b;
                    float* t4 = &geom[4 * ((size_t)n_padded + 2u * n_blocks + 2u * (i / RT_Q_STREAM_TILE))];
                    if (!(a >= t4[k])) t4[k] = a;
                    if (!(b <= t4[4 + k])) t4[4 + k] = b;
                }
            }
            x[3] = rr;
            float param = sp.param;
            if (sp.material == RT_MAT_REFLECT) param = param < 1.0f ? param : 1.0f;   // library.cu:144
            s[0] = sp.center[0]; s[1] = sp.center[1]; s[2] = sp.center[2]; s[3] = sp.radius;
            s[4] = sp.albedo[0]; s[5] = sp.albedo[1]; s[6] = sp.albedo[2]; s[7] = param;
            mat[i] = sp.material;
            if (sp.material == RT_MAT_DIELECTRIC) {
                // constants of the book's dielectric, in the single-precision operations of the kernel's former
                // inline code (and of the checker): 1/ior, and Schlick's r0 for a ray entering (ri = 1/ior) and leaving (ri = ior)
                float* gt = &shade[8 * (size_t)n_padded + 4 * (size_t)i];
                const float inv = 1.0f / param;
                float r0 = (1.0f - inv) / (1.0f + inv);
                gt[0] = inv;
                gt[1] = r0 * r0;
                r0 = (1.0f - param) / (1.0f + param);
                gt[2] = r0 * r0;
            }
        } else {
            // padding: m = -inf makes the filter value -inf, never a candidate
            g[0] = g[1] = g[2] = 0.0f; g[3] = -INFINITY;
            x[0] = x[1] = x[2] = 0.0f; x[3] = -INFINITY;
            s[0] = s[1] = s[2] = 0.0f; s[3] = 1.0f;
            s[4] = s[5] = s[6] = s[7] = 0.0f;
            mat[i] = RT_MAT_DIFFUSE;
        }
    }
    // ... and behind the tiles' boxes a ball per block that holds the block's box, {cb, rb (1 + 2^-7)}: what the pixel
    // units' lists of a streamed scene test before they look at a block's spheres (unit_list in rt_kernels.hip).  cb is
    // the box's centre as rounded to single precision, rb half the diagonal plus that rounding, rounded up.
    for (uint32_t b = 0; b < n_blocks; b++) {
        const float* lo4 = &geom[4 * ((size_t)n_padded + 2u * b)];
        float* ball = &geom[4 * ((size_t)n_padded + 2u * n_blocks + 2u * n_tiles + b)];
        double d2 = 0.0, c1 = 0.0;
        bool finite = true;
        for (int k = 0; k < 3; k++) {
            finite = finite && std::isfinite(lo4[k]) && std::isfinite(lo4[4 + k]) && lo4[k] <= lo4[4 + k];
            const double mid = 0.5 * ((double)lo4[k] + (double)lo4[4 + k]), ext = (double)lo4[4 + k] - (double)lo4[k];
            ball[k] = (float)mid;
            d2 += ext * ext;
            c1 += std::fabs(mid);
        }
        const double rb = (0.5 * std::sqrt(d2) * (1.0 + 0x1p-40) + 0x1p-22 * c1) * (1.0 + 0x1p-7);
        ball[3] = std::nextafter((float)rb, INFINITY);
        if (!finite || !std::isfinite(ball[3])) { ball[0] = ball[1] = ball[2] = 0.0f; ball[3] = INFINITY; }
    }
    size_t bytes = (size_t)n_padded * 16u;
    const size_t geom_bytes = bytes + ((size_t)n_blocks + n_tiles) * 32u + (size_t)n_blocks * 16u;
    // One allocation, the arrays at offsets that follow from n_padded alone (SceneAt in rt_kernels.hip finds them from
    // the first address): filter data with the boxes and balls, exact-test data, shading data (3 float4 per sphere),
    // the launch's camera-ray terms, then the material tags and the caller's index of each device sphere.
    void* old_scene = c->d_geom;
    c->d_geom = c->d_exact = c->d_shade = c->d_mat = c->d_camterms = nullptr;
    c->have_scene = false;
    if (old_scene) { RT_HIP(c, hipFree(old_scene)); }
    RT_HIP(c, hipMalloc(&c->d_geom, geom_bytes + 5 * bytes + (size_t)n_padded * 8u));
    c->d_exact = (char*)c->d_geom + geom_bytes;
    c->d_shade = (char*)c->d_exact + bytes;
    c->d_camterms = (char*)c->d_shade + 3 * bytes;
    c->d_mat = (char*)c->d_camterms + bytes;
    RT_HIP(c, hipMemcpy(c->d_geom, geom.data(), geom_bytes, hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(c->d_exact, exact.data(), bytes, hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(c->d_shade, shade.data(), 3 * bytes, hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(c->d_mat, mat.data(), (size_t)n_padded * 8u, hipMemcpyHostToDevice));
    for (int k = 0; k < 3; k++) c->translate[k] = T[k];
    c->scan_scale = sigma;
    c->scene_radius = scene_radius;
    c->n_spheres = n;
    c->n_padded = n_padded;
    c->n_blocks = sorted ? n_blocks : 0u;
    c->have_scene = true;
    return 0;
}

int rt_render_rows(rt_ctx* c, const RtCamera* cam, const RtParams* p, uint32_t row0, uint32_t rows,
                   uint8_t* rgba_out, float* linear_out, int32_t* hit_out, uint64_t* rays_out) {
    int rc = check_common(c, cam, p);
    if (rc) return rc;
    if ((rc = check_render(c, p))) return rc;
    if (!rgba_out) return fail(c, RT_ERR_INVALID, "rgba_out is NULL");
    if (rows == 0 || row0 >= cam->image_height || rows > cam->image_height - row0)
        return fail(c, RT_ERR_INVALID, "row range outside the image");
    RT_HIP(c, hipSetDevice(c->device));
    size_t px = (size_t)rows * cam->image_width;
    if ((rc = ensure(c, &c->d_rgba, &c->cap_rgba, px * 4))) return rc;
    if (linear_out && (rc = ensure(c, &c->d_linear, &c->cap_linear, px * 16))) return rc;
    RtLaunch L = make_launch(c, cam, p, row0, rows, rows, rows, true);
    if ((rc = launch_render(c, cam, p, L, c->d_rgba, linear_out ? c->d_linear : nullptr, rays_out, c->d_counter))) return rc;
    RT_HIP(c, hipMemcpy(rgba_out, c->d_rgba, px * 4, hipMemcpyDeviceToHost));      // memcpy_dtoh, ray_tracing.rs:206-208
    if (linear_out) RT_HIP(c, hipMemcpy(linear_out, c->d_linear, px * 16, hipMemcpyDeviceToHost));
    if (hit_out) return rt_primary_hits(c, cam, p, row0, rows, hit_out);
    return 0;
}

// rt_render_tiles_device and its allocation-free twin: same checks, same launch, different counter block
static int render_tiles_into(rt_ctx* c, const RtCamera* cam, const RtParams* p, uint32_t tile_rows, uint32_t first_tile,
                             uint32_t tile_stride, uint32_t n_tiles, void* d_rgba, void* d_linear, uint64_t* rays_out,
                             void* counters) {
    if (!d_rgba) return fail(c, RT_ERR_INVALID, "d_rgba is NULL");
    if (tile_rows == 0 || tile_rows % 8u != 0) return fail(c, RT_ERR_INVALID, "tile_rows must be a positive multiple of 8");
    if (tile_stride == 0) return fail(c, RT_ERR_INVALID, "tile_stride must be >= 1");
    if ((uint64_t)n_tiles * tile_rows >= (1ull << 31)) return fail(c, RT_ERR_INVALID, "too many tile rows");
    if (n_tiles && (uint64_t)first_tile * tile_rows >= cam->image_height) return fail(c, RT_ERR_INVALID, "first_tile lies below the image");
    if ((uint64_t)tile_stride * tile_rows >= (1ull << 31)) return fail(c, RT_ERR_INVALID, "tile_stride * tile_rows must be below 2^31");
    RT_HIP(c, hipSetDevice(c->device));
    if (n_tiles == 0) { if (rays_out) *rays_out = 0; c->last_ms = 0.0f; return 0; }
    // the allocation-free entry point keeps the fold on the kernel that needs no workspace
    const bool may_allocate = counters == c->d_counter;
    RtLaunch L = make_launch(c, cam, p, first_tile * tile_rows, n_tiles * tile_rows, tile_rows, tile_stride * tile_rows, may_allocate);
    return launch_render(c, cam, p, L, d_rgba, d_linear, rays_out, counters, may_allocate);
}

int rt_render_tiles_device(rt_ctx* c, const RtCamera* cam, const RtParams* p, uint32_t tile_rows,
                           uint32_t first_tile, uint32_t tile_stride, uint32_t n_tiles,
                           void* d_rgba, void* d_linear, uint64_t* rays_out) {
    int rc = check_common(c, cam, p);
    if (rc) return rc;
    if ((rc = check_render(c, p))) return rc;
    return render_tiles_into(c, cam, p, tile_rows, first_tile, tile_stride, n_tiles, d_rgba, d_linear, rays_out, c->d_counter);
}

// No hipMalloc / hipFree / allocating routine on this path: argument checks, one occupancy query, hipMemsetAsync
// of the counter block, the launch between two event records, a stream synchronisation and an 8-byte copy of the
// segment count.  The counter block is the library's static device array (address looked up per GPU in rt_create).
int rt_render_tiles_device_static(rt_ctx* c, const RtCamera* cam, const RtParams* p, uint32_t tile_rows,
                                  uint32_t first_tile, uint32_t tile_stride, uint32_t n_tiles,
                                  void* d_rgba, void* d_linear, uint64_t* rays_out) {
    int rc = check_common(c, cam, p);
    if (rc) return rc;
    if ((rc = check_render(c, p))) return rc;
    if (!c->d_scratch)
        return fail(c, c->scratch_rc ? c->scratch_rc : RT_ERR_INVALID,
                    std::string("static scratch array unavailable on this GPU: ") +
                        (c->scratch_rc ? hipGetErrorName((hipError_t)c->scratch_rc) : "not looked up"));
    // The one size that matters here: the counter block the loaded code object asks for (its rt_counter_bytes;
    // code objects can be replaced at run time with rt_load_module) must fit the static array.  It does not grow
    // with the frame, the scene or the sample count.
    if (c->counter_need > RT_SCRATCH_BYTES)
        return fail(c, RT_ERR_INVALID, "rt_render_tiles_device_static: the loaded code object needs " + std::to_string(c->counter_need) +
                                           " bytes of counters, the static scratch array holds " + std::to_string(RT_SCRATCH_BYTES));
    return render_tiles_into(c, cam, p, tile_rows, first_tile, tile_stride, n_tiles, d_rgba, d_linear, rays_out, c->d_scratch);
}

int rt_primary_hits(rt_ctx* c, const RtCamera* cam, const RtParams* p, uint32_t row0, uint32_t rows, int32_t* hit_out) {
    int rc = check_common(c, cam, p);
    if (rc) return rc;
    if (!hit_out) return fail(c, RT_ERR_INVALID, "hit_out is NULL");
    if (rows == 0 || row0 >= cam->image_height || rows > cam->image_height - row0)
        return fail(c, RT_ERR_INVALID, "row range outside the image");
    RT_HIP(c, hipSetDevice(c->device));
    size_t px = (size_t)rows * cam->image_width;
    if ((rc = ensure(c, &c->d_hits, &c->cap_hits, px * 4))) return rc;
    RtLaunch L = make_launch(c, cam, p, row0, rows, rows, rows);
    if ((rc = launch_hits(c, cam, p, L, c->d_hits))) return rc;
    RT_HIP(c, hipMemcpy(hit_out, c->d_hits, px * 4, hipMemcpyDeviceToHost));
    return 0;
}

int rt_show_random(rt_ctx* c, uint32_t width, uint32_t height, uint64_t seed, int norm, uint8_t* rgba_out) {
    if (!c) return RT_ERR_INVALID;
    if (!c->module) return fail(c, RT_ERR_NO_MODULE, "no code object loaded");
    if (!rgba_out || !width || !height || (uint64_t)width * height >= (1ull << 31))
        return fail(c, RT_ERR_INVALID, "bad frame size or NULL output");
    RT_HIP(c, hipSetDevice(c->device));
    uint32_t n = width * height;
    int rc = ensure(c, &c->d_rgba, &c->cap_rgba, (size_t)n * 4);
    if (rc) return rc;
    uint32_t lo = (uint32_t)(seed & 0xffffffffu), hi = (uint32_t)(seed >> 32), nm = norm ? 1u : 0u;
    void* args[] = {&c->d_rgba, &n, &lo, &hi, &nm};
    RT_HIP(c, hipModuleLaunchKernel(c->f_random, (n + 255u) / 256u, 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(rgba_out, c->d_rgba, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

float rt_last_kernel_ms(const rt_ctx* c) { return c ? c->last_ms : 0.0f; }

const char* rt_last_kernel_name(const rt_ctx* c) { return c ? c->last_kernel : ""; }

int rt_debug_counters(rt_ctx* c, uint64_t out[16]) {
    if (!c || !out) return RT_ERR_INVALID;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipMemcpy(out, c->last_counters ? c->last_counters : c->d_counter, RT_COUNTER_BYTES, hipMemcpyDeviceToHost));
    return 0;
}

int rt_device_info(const rt_ctx* c, char* name, size_t name_size, int* compute_units, int* clock_khz) {
    if (!c) return RT_ERR_INVALID;
    if (name && name_size) std::snprintf(name, name_size, "%s", c->name);
    if (compute_units) *compute_units = c->compute_units;
    if (clock_khz) *clock_khz = c->clock_khz;
    return 0;
}

int rt_comm_id(uint8_t id[RT_COMM_ID_BYTES]) {
    if (!id) return RT_ERR_INVALID;
    Rccl* r = rccl();
    if (!r->handle) return fail(nullptr, RT_ERR_COMM, r->error);
    ncclUniqueId u;
    static_assert(sizeof u == RT_COMM_ID_BYTES, "RT_COMM_ID_BYTES must be the size of ncclUniqueId");
    RT_RCCL(nullptr, r->GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return 0;
}

int rt_comm_init(rt_ctx* c, const uint8_t id[RT_COMM_ID_BYTES], uint32_t world, uint32_t rank) {
    if (!c) return RT_ERR_INVALID;
    if (!id || world == 0 || rank >= world) return fail(c, RT_ERR_INVALID, "rt_comm_init: need an id and rank < world");
    if (c->comm) return fail(c, RT_ERR_INVALID, "rt_comm_init: this context already has a communicator");
    Rccl* r = rccl();
    if (!r->handle) return fail(c, RT_ERR_COMM, r->error);
    RT_HIP(c, hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    if (!c->d_status) {
        RT_HIP(c, hipMalloc(&c->d_status, 12));
        const int32_t init[3] = {0, 0, 1};
        RT_HIP(c, hipMemcpy(c->d_status, init, sizeof init, hipMemcpyHostToDevice));
    }
    RT_RCCL(c, r->CommInitRank(&c->comm, (int)world, u, (int)rank));
    c->world = world;
    c->rank = rank;
    return 0;
}

int rt_assemble_tiles(rt_ctx* c, const void* d_gathered, uint32_t width, uint32_t height, uint32_t tile_rows,
                      uint32_t world, void* d_frame) {
    if (!c) return RT_ERR_INVALID;
    if (!c->module) return fail(c, RT_ERR_NO_MODULE, "no code object loaded");
    if (!d_gathered || !d_frame || !width || !height || !world || tile_rows == 0 || tile_rows % 8u != 0 ||
        (uint64_t)width * height >= (1ull << 31))
        return fail(c, RT_ERR_INVALID, "rt_assemble_tiles: NULL buffer, empty frame, or tile_rows not a positive multiple of 8");
    RT_HIP(c, hipSetDevice(c->device));
    const uint32_t rows_per_rank = tiles_of_rank(height, tile_rows, 0, world) * tile_rows;
    int rc = launch_assemble(c, const_cast<void*>(d_gathered), d_frame, width, height, tile_rows, world, rows_per_rank);
    if (rc) return rc;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int rt_render_frame_sharded(rt_ctx* c, const RtCamera* cam, const RtParams* p, uint32_t tile_rows,
                            uint8_t* frame_out, uint64_t* rays_out) {
    // Argument errors are the same on every rank (same camera and parameters by contract), so returning early
    // on them leaves no peer waiting.  Everything that can fail on one rank only (allocation, launch) happens
    // before the ranks exchange a status word; the tile buffers travel only if every rank is ready.
    int rc = check_common(c, cam, p);
    if (rc) return rc;
    if ((rc = check_render(c, p))) return rc;
    if (tile_rows == 0 || tile_rows % 8u != 0) return fail(c, RT_ERR_INVALID, "tile_rows must be a positive multiple of 8");
    if (c->world > 1 && !c->comm) return fail(c, RT_ERR_COMM, "this context's communicator was given up after a failed exchange; create a new context and communicator");
    const uint32_t W = cam->image_width, H = cam->image_height, world = c->world, rank = c->rank;
    const uint32_t rows_per_rank = tiles_of_rank(H, tile_rows, 0, world) * tile_rows;   // rank 0 owns the most tiles
    const size_t part = (size_t)rows_per_rank * W * 4;                                  // bytes every rank contributes
    const size_t frame_bytes = (size_t)W * H * 4;
    uint64_t rays = 0;
    auto local_part = [&]() -> int {
        if (rank == 0 && !frame_out) return fail(c, RT_ERR_INVALID, "frame_out is NULL on rank 0");
        RT_HIP(c, hipSetDevice(c->device));
        // rank 0 keeps all ranks' buffers side by side and renders into the first; the others hold their own only
        int r = ensure(c, &c->d_tiles, &c->cap_tiles, rank == 0 ? part * world : part);
        if (r) return r;
        if (rank == 0 && (r = ensure(c, &c->d_frame, &c->cap_frame, frame_bytes))) return r;
        const uint32_t mine = tiles_of_rank(H, tile_rows, rank, world);
        if (mine) {
            RtLaunch L = make_launch(c, cam, p, rank * tile_rows, mine * tile_rows, tile_rows, world * tile_rows, true);
            if ((r = launch_render(c, cam, p, L, c->d_tiles, nullptr, &rays, c->d_counter))) return r;
        }
        return 0;
    };
    const int local_rc = local_part();
    const std::string local_err = c->err;
    if (c->comm) {
        // one 4-byte all-reduce (max) of the ranks' status: a rank that failed above still takes part, so nobody
        // blocks in the exchange below waiting for a peer that has already returned
        Rccl* r = rccl();
        const int32_t mine_failed = local_rc ? 1 : 0;
        int32_t any_failed = 1;
        char* status = static_cast<char*>(c->d_status);
        // A rank that cannot even upload its status word still takes part, sending the constant 1 kept in word 2:
        // its peers must not be left waiting in the all-reduce.
        const void* send = status;
        int upload_rc = 0;
        hipError_t e = hipMemcpyAsync(c->d_status, &mine_failed, 4, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { upload_rc = hip_fail(c, e, "hipMemcpyAsync(status)"); send = status + 8; }
        const std::string upload_err = c->err;
        // If the exchange itself fails on this rank, nobody can tell any more who goes on to send tiles: the
        // communicator is given up (ncclCommAbort releases operations of this rank that peers may be waiting in).
        auto give_up = [&](int code) {
            if (r->CommAbort) (void)r->CommAbort(c->comm); else (void)r->CommDestroy(c->comm);
            c->comm = nullptr;
            return code;
        };
        ncclResult_t ar = r->AllReduce(send, status + 4, 1, ncclInt32, ncclMax, c->comm, c->stream);
        if (ar != ncclSuccess) return give_up(rccl_fail(c, ar, "ncclAllReduce(status)"));
        if ((e = hipMemcpyAsync(&any_failed, status + 4, 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(c->stream)) != hipSuccess)
            return give_up(hip_fail(c, e, "status exchange"));
        if (upload_rc) return fail(c, upload_rc, upload_err);
        if (local_rc) return fail(c, local_rc, local_err);
        if (any_failed) return fail(c, RT_ERR_COMM, "rt_render_frame_sharded: another rank failed before the exchange; no tiles were sent");
    } else if (local_rc) {
        return local_rc;
    }
    if (rays_out) *rays_out = rays;
    if (world > 1) {
        // the group is closed on every path, also after a failed send or receive
        Rccl* r = rccl();
        ncclResult_t first = r->GroupStart();
        const char* what = "ncclGroupStart";
        if (first == ncclSuccess) {
            if (rank == 0) {
                for (uint32_t peer = 1; peer < world && first == ncclSuccess; peer++) {
                    first = r->Recv(static_cast<char*>(c->d_tiles) + part * peer, part, ncclUint8, (int)peer, c->comm, c->stream);
                    what = "ncclRecv";
                }
            } else {
                first = r->Send(c->d_tiles, part, ncclUint8, 0, c->comm, c->stream);
                what = "ncclSend";
            }
            const ncclResult_t end = r->GroupEnd();
            if (first == ncclSuccess && end != ncclSuccess) { first = end; what = "ncclGroupEnd"; }
        }
        if (first != ncclSuccess) return rccl_fail(c, first, what);
    }
    if (rank == 0) {
        if ((rc = launch_assemble(c, c->d_tiles, c->d_frame, W, H, tile_rows, world, rows_per_rank))) return rc;
        RT_HIP(c, hipMemcpyAsync(frame_out, c->d_frame, frame_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int rt_debug_fold_workspace_limit(rt_ctx* c, uint64_t bytes) {
    if (!c) return RT_ERR_INVALID;
    c->fold_limit = bytes ? (size_t)bytes : (size_t)8 << 30;
    return 0;
}

int rt_debug_cull(rt_ctx* c, int unit_lists, int block_boxes) {
    if (!c) return RT_ERR_INVALID;
    c->cull_lists = unit_lists != 0;
    c->cull_boxes = block_boxes != 0;
    return 0;
}

int rt_debug_schedule(rt_ctx* c, uint32_t unit_w, uint32_t unit_h, int no_queue, int scheduler) {
    if (!c) return RT_ERR_INVALID;
    auto pow2 = [](uint32_t v) { return v && !(v & (v - 1u)); };
    if (unit_w || unit_h) {
        if (!pow2(unit_w) || !pow2(unit_h) || unit_w * unit_h > 64u)
            return fail(c, RT_ERR_INVALID, "rt_debug_schedule: unit sides must be powers of two with at most 64 pixels per unit");
    }
    c->dbg_unit_w = unit_w;
    c->dbg_unit_h = unit_h;
    if (scheduler != RT_SCHED_AUTO && scheduler != RT_SCHED_WAVE && scheduler != RT_SCHED_POOL)
        return fail(c, RT_ERR_INVALID, "rt_debug_schedule: scheduler must be RT_SCHED_AUTO, RT_SCHED_WAVE or RT_SCHED_POOL");
    c->dbg_no_queue = no_queue != 0;
    c->dbg_scheduler = scheduler;
    return 0;
}

}  // extern "C"
