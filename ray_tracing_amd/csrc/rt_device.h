// rt_device.h -- kernel-side layout shared by the gfx950 code object (rt_kernels.hip) and the
// host launcher (rt_hip.cpp).  Not part of the public interface.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <stdint.h>
#include "../../include/rt_types.h"

// Second kernel argument, by value (the camera block is the first by-value argument, as in the
// reference's `render(rng_state, frame, Camera cam, GuiState gui)`, ray.cu:92).
typedef struct RtLaunch {
    uint32_t n_spheres;        // spheres in the scene
    uint32_t n_padded;         // n_spheres rounded up to RT_SCAN_PAD (padding never hits)
    uint32_t samples_per_pixel;
    uint32_t sample2_per_pixel;
    uint32_t max_depth;
    uint32_t flags;
    uint32_t seed_lo, seed_hi;
    float    t_min, t_max;
    uint32_t width, height;    // full image
    uint32_t row_base;         // global row of local row 0
    uint32_t rows_local;       // rows this launch renders (compact output buffer height)
    uint32_t tile_rows;        // local rows per row tile
    uint32_t tile_stride_rows; // global row distance between consecutive row tiles of this launch
    uint32_t tiles_x;          // 8x8 wave tiles per image row = ceil(width / 8)          (hit-map kernel)
    uint32_t n_wave_tiles;     // tiles_x * ceil(rows_local / 8)
    // render kernels: wavefronts take units of unit_w x unit_h pixels, samples cut into runs of `chunk`
    uint32_t unit_w, unit_h;   // powers of two, unit_w * unit_h <= 64; unit_h divides tile_rows
    uint32_t unit_shift;       // log2(unit_w)
    uint32_t unit_pshift;      // log2(unit_w * unit_h)
    uint32_t units_x;          // ceil(width / unit_w)
    uint32_t n_units;          // units_x * ceil(rows_local / unit_h)
    uint32_t chunk;            // samples per work item (= samples_per_pixel when n_chunks == 1)
    uint32_t n_chunks;         // ceil(samples_per_pixel / chunk); 1 = the reference's fp32 fold
    float    translate[3];     // T of the scan filter: component-wise median of the sphere centres
    uint32_t fold;             // queue kernels: 1 = finished samples go to the workspace and are added in sample order (fp32 fold)
    uint32_t pool_slots;       // rt_render_pool: open units per workgroup (LDS pool)
    uint32_t units_bottom_up;  // 1: the wavefronts take the launch's pixel units from the last row upwards
    float    scan_scale;       // sigma of the scan filter: a power of two, sigma * |c - o| <= 1/2 for origins on the scene's spheres
    uint32_t sorted;           // 1: the device sphere list is sorted in space; the caller's indices follow the material tags
    uint32_t scan_blocks;      // blocks of 32 spheres with a bounding box (two float4 each after the n_padded filter entries); 0: none
    uint32_t ulist_cap;        // camera-ray-batch kernels: entries of a pixel unit's sphere list (even, <= 32: one mask of the scan; 0: no lists)
    float    unit_reach;       // how far a camera ray's direction can be from the one through the centre of its unit's footprint
} RtLaunch;

// Launch-wide counter block of the render kernels, device memory, cleared before every launch: sixteen 64-bit
// words; [0] path segments traced, [1..7] and [9..15] diagnostics of -DRT_STATS builds, byte 64: the 32-bit
// counter the wavefronts take pixel units from.
// The code object states the size it was built with in its global word `rt_counter_bytes`; the launcher sizes
// (or, for the static scratch array, checks) the block by that word.
#ifndef RT_COUNTER_BYTES
#define RT_COUNTER_BYTES 128
#endif

// spheres per unrolled scan step (8 measured best: 4 is 3 % slower, 16 costs a wave of occupancy and 2 %)
#ifndef RT_SCAN_UNROLL
#define RT_SCAN_UNROLL 8u
#endif
// ... and in the builds with block boxes, where the dense scan is the lesser part and registers are short (8 there:
// 12 to 56 more bytes of scratch per lane, book frame 311 -> 326 ms, mirror hall 331 -> 350, 16,385 spheres 70.5 -> 74)
#ifndef RT_SCAN_UNROLL_BOXES
#define RT_SCAN_UNROLL_BOXES 4u
#endif
// ... and in the camera-ray scan
#ifndef RT_SCAN_UNROLL_PRIMARY
#define RT_SCAN_UNROLL_PRIMARY 4u
#endif
// sphere lists are padded to a multiple of this (the scans end with a step of 4 and a step of 2 spheres)
#define RT_SCAN_PAD 2u
// Spheres staged in LDS at a time: 1024 x 32 B = 32 KiB.  Scenes up to this size are staged once per
// workgroup; larger ones are streamed through the same buffer tile by tile.  A bigger tile means fewer
// barriers per path segment but fewer workgroups per CU: on the 16,385-sphere scene 512 and 1024 run at
// 232 Mrays/s, 2048 (two workgroups per CU) at 169.
#ifndef RT_TILE_SPHERES
#define RT_TILE_SPHERES 1024u
#endif
// Candidate slots per lane (u16 each, in LDS); a wavefront drains its lists as soon as one lane's list
// is full (checked after every push, so a list can never overflow).
#ifndef RT_CAND_CAP
#define RT_CAND_CAP 16u
#endif
// Kernel for sample_chunk = 1 (camera rays in batches): candidate slots per lane, 32-bit words per queued ray;
// spheres per LDS tile when that kernel streams a longer list (a multiple of 32: a tile is whole blocks; at most 512:
// a wavefront tests the boxes of a tile's blocks with 16 lanes).  Measured on the 16,385-sphere scene, with a box per
// tile and per block and the launcher's workgroup size of the time: 256-thread workgroups 256: 226 ms, 512: 207 ms,
// 896: 205 ms; later, workgroups of two wavefronts 512: 87 ms, 384: 88, 448: 93; workgroups of ONE wavefront (nothing
// to wait for at the tile barriers, and small tiles let 18 of them share a compute unit's LDS) 512: 94 ms, 256: 73 ms, 128 (20
// resident, twice the tile boxes to vote on): 88 ms.
// Without boxes, when every tile was scanned: 512: 1,265 ms, 1024: 1,341 ms, 2048: 1,748 ms.
#ifndef RT_Q_STREAM_TILE
#define RT_Q_STREAM_TILE 256u
#endif
// A block of 32 spheres reached by at most this many lanes of a wavefront is scanned lanes-over-spheres, one reaching ray at
// a time (scan_tile); beyond it the 64 rays scan the block's spheres side by side.
#ifndef RT_SPARSE_MAX
#define RT_SPARSE_MAX 16u
#endif
// ... and a streamed tile reached by at most this many lanes has the boxes of its blocks tested lanes-over-blocks
#ifndef RT_SPARSE_TILE_MAX
#define RT_SPARSE_TILE_MAX 12u
#endif
// waves per SIMD the streaming camera-ray-batch kernels are built for (16,385-sphere frame: 5 -- 96 registers, 72 bytes
// of scratch per lane -- 73 ms; 4 -- 128 registers, 12 bytes -- 77 ms: the kernel waits on memory and LDS, a resident
// wavefront more is worth more than the spills cost while they stay in L2)
#ifndef RT_STREAM_WAVES
#define RT_STREAM_WAVES 5
#endif
#define RT_Q_CAND_CAP 8u
#define RT_Q_WORDS 8u
// LDS bytes of a pixel unit's sphere list of `cap` entries: a float4 of camera-ray terms and a u16 index each
#define RT_UL_BYTES(cap) (((cap) * 18u + 15u) & ~15u)

#endif
