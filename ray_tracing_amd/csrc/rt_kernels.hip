// rt_kernels.hip -- gfx950 code object of the path tracer: the sphere list streamed from LDS (every lane
// of a wavefront reads the same float4 per step), counter-based PCG32 streams, fp32 register
// accumulation, gamma + RGBA8 store.  The render kernels share the code below:
//   rt_render_queue         every sample its own work item, scene within one LDS tile (the fast path)
//   rt_render_queue_boxes   the same with a bounding box per block of 32 spheres of a list sorted in space: a bounce
//                           ray's scan skips the blocks no lane reaches, and scans a block few lanes reach for those
//                           lanes only, lanes over spheres (scan_tile, sparse_block)
//   rt_render_pool, rt_render_pool_boxes, rt_render_pool_plain   the same with the pixel units pooled per workgroup
//   rt_render_queue_stream  the same for longer sphere lists, streamed tile by tile (scan_tile_stream)
//   rt_render_queue_fold, rt_render_pool_fold, rt_render_queue_stream_fold   adding each pixel's samples in sample order
//   rt_render               work items of `chunk` samples, or one lane per pixel in the reference's fold order
//   rt_render_stream        the same for longer sphere lists
// and rt_camera_terms runs before the camera-ray-batch kernels: the camera-ray scan's four numbers per sphere for the
// launch's camera.
// Built with
//   hipcc --offload-arch=gfx950 --genco -O3 -ffp-contract=off
// and loaded with hipModuleLoadData by rt_hip.cpp.  No MFMA: the work is divergent scalar fp32.
//
// Arithmetic contract: every floating-point operation below is written out (explicit fmaf, IEEE
// divide and sqrt, no compiler contraction), in the order of the single-precision CPU checker, so
// radiance and hit indices reproduce bit for bit.  The op order follows the reference's
// gpu-implementation/src/cuda/{floatN_helper,library,ray}.cu as nvcc contracts it; citations below
// are relative to gpu-implementation/src/cuda/ unless they name lib.rs (cpu-implementation/src/lib.rs).
#include <hip/hip_runtime.h>
#include "rt_device.h"

#define RT_DEV __device__ __forceinline__

// size of the launch-wide counter block these kernels were built for (read by the launcher when the object is loaded)
extern "C" __device__ __attribute__((used)) const uint32_t rt_counter_bytes = RT_COUNTER_BYTES;
// ... and the tile size of its streaming kernels: the scene's layout in device memory follows from it (SceneAt), so the
// launcher refuses an object built for another one.
extern "C" __device__ __attribute__((used)) const uint32_t rt_stream_tile = RT_Q_STREAM_TILE;

struct V3 { float x, y, z; };
RT_DEV V3 v3(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
RT_DEV V3 vsub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 vadd(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 vscale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
// a.x*b.x + a.y*b.y + a.z*b.z (floatN_helper.cu:109-111) as an fma chain
RT_DEV float vdot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// v * (1/|v|): stands in for rnorm3df (floatN_helper.cu:120-122); cgmath's normalize (lib.rs:44,203)
#ifdef RT_ABL_VUNIT   // ablation (changes the image): what the IEEE square root and divide of a normalisation cost
RT_DEV V3 vunit(V3 v) { float inv = __builtin_amdgcn_rsqf(vdot(v, v)); return vscale(v, inv); }
#else
RT_DEV V3 vunit(V3 v) { float inv = 1.0f / sqrtf(vdot(v, v)); return vscale(v, inv); }
#endif
RT_DEV V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

// ---------------------------------------------------------------- diagnostics
// Built with -DRT_STATS the kernels also count loop trips and shader cycles per section, summed over the
// launch into the counter block (64-bit words).  [1] scan passes per wavefront, [2] candidates listed (per
// lane), [3] exact-test rounds per wavefront, [4] decode rounds per wavefront, [5] unit-vector tries per lane,
// [6] drain calls per wavefront, [7] unit-vector rounds per wavefront; cycles a wavefront spent (s_memtime,
// including the time its SIMD ran other wavefronts) in [9] the scan filter, [10] decode, [11] exact tests,
// [12] hit record + scatter, [13] sky / next sample / run flush, [14] unit and item hand-out, [15] the rest
// of the loop.  Without RT_STATS the calls compile to nothing.
struct Stats {
#ifdef RT_STATS
    uint32_t v[8];
    unsigned long long cyc[7], last;
    RT_DEV Stats() { for (int i = 0; i < 8; i++) v[i] = 0u; for (int i = 0; i < 7; i++) cyc[i] = 0ull; last = __builtin_amdgcn_s_memtime(); }
    RT_DEV void raw(int which, uint32_t n) { v[which] += n; }
#ifdef RT_STATS_REACH   // words 2..7 count block boxes instead: [2] tested, then reached by 1-2, 3-4, 5-8, 9-16, 17-64 lanes
    RT_DEV void add(int, uint32_t) {}
    RT_DEV void wave(int which) { if (which == 1) reach(1); }
    RT_DEV void reach(int which) {
#else
    RT_DEV void add(int which, uint32_t n) { v[which] += n; }
    RT_DEV void reach(int) {}
    // counted once per wavefront: by the first lane that is active at the call
    RT_DEV void wave(int which) {
#endif
        unsigned long long m = __ballot(1);
        if ((uint32_t)(__ffsll((long long)m) - 1) == (threadIdx.x & 63u)) v[which] += 1u;
    }
    // cycles since the previous lap go to section `which` (9..15)
    RT_DEV void lap(int which) {
        unsigned long long now = __builtin_amdgcn_s_memtime();
        cyc[which - 9] += now - last;
        last = now;
    }
#else
    RT_DEV void add(int, uint32_t) {}
    RT_DEV void raw(int, uint32_t) {}
    RT_DEV void wave(int) {}
    RT_DEV void reach(int) {}
    RT_DEV void lap(int) {}
#endif
};

// ---------------------------------------------------------------- RNG
// PCG32 XSH-RR (library.cu:7-16), one stream per (seed, pixel, sample, k): k = 0 feeds the camera
// jitter, k = j+1 the scatter that ends path segment j.  Seeding is pcg32_srandom_r(key, pixel) with
// key = splitmix64 finaliser of seed ^ (sample << 32 | k).
struct Rng { uint64_t state, inc; };
RT_DEV uint32_t pcg_next(Rng& g) {
    uint64_t old = g.state;
    g.state = old * 6364136223846793005ULL + g.inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31u));
}
RT_DEV uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
RT_DEV Rng stream_init(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k) {
    Rng g;
#ifdef RT_ABL_STREAM  // ablation (changes the image): what the three 64-bit multiplies of a stream start cost
    g.inc = ((uint64_t)pixel << 1u) | 1u;
    g.state = (seed ^ (((uint64_t)sample << 32) | (uint64_t)k)) + g.inc;
    return g;
#endif
    uint64_t key = mix64(seed ^ (((uint64_t)sample << 32) | (uint64_t)k));
    g.inc = ((uint64_t)pixel << 1u) | 1u;
    g.state = g.inc + key;                       // state 0 stepped once is inc; then += initstate
    g.state = g.state * 6364136223846793005ULL + g.inc;
    return g;
}
// top 24 bits -> [0,1), exact (the reference's ldexpf(u,-32), library.cu:18-24, can round to 1.0)
RT_DEV float rand01(Rng& g) { return (float)(pcg_next(g) >> 8) * 0x1p-24f; }

// Rejection in (-1,1)^3, accept p.p <= 1, normalise (lib.rs:31-47); at most 20 tries, then (1,0,0)
// (library.cu:25-31).  One try takes ONE generator output a and cuts three coordinates out of its 32 bits:
// x = a >> 21 and y = (a >> 10) & 0x7ff (11 bits, value (k + 1/2) 2^-10 - 1), z = a & 0x3ff (10 bits,
// (k + 1/2) 2^-9 - 1): cell centres of a regular grid over the cube, symmetric about the origin and never the zero
// vector; all exact.  (One 64-bit multiply per try: integer multiplies run at a quarter of the fp32 rate, and the
// lanes of a wavefront wait for the one with the most tries.)
RT_DEV V3 random_unit_vector(Rng& g, Stats& st) {
    V3 p = v3(1.0f, 0.0f, 0.0f);
    bool done = false;
    for (int i = 0; i < 20 && !done; i++) {
        st.add(5, 1u);
        st.wave(7);
        const uint32_t a = pcg_next(g);
        float x = fmaf((float)(a >> 21), 0x1p-10f, -1.0f + 0x1p-11f);
        float y = fmaf((float)((a >> 10) & 0x7ffu), 0x1p-10f, -1.0f + 0x1p-11f);
        float z = fmaf((float)(a & 0x3ffu), 0x1p-9f, -1.0f + 0x1p-10f);
        V3 q = v3(x, y, z);
        if (vdot(q, q) <= 1.0f) { p = q; done = true; }
    }
    // normalised once, after the loop, for all lanes together; the fallback (1,0,0) is its own unit vector
    return vunit(p);
}

// ---------------------------------------------------------------- closest hit
struct Best { float t; int idx; };
#ifdef RT_BOXCHECK   // diagnostics build: see scan_tile
__device__ unsigned long long* rt_dbg_ptr = nullptr;
#define RT_DBG_SET(p) rt_dbg_ptr = (p)
#else
#define RT_DBG_SET(p)
#endif

// Exact acceptance of Sphere::hit (library.cu:205-218, lib.rs:103-117) for one sphere against the
// current upper bound.  Ties keep the lower index -- the CALLER's index: the device list is sorted in space, orig[] maps
// back, and two spheres reporting the very same root (a small sphere resting on the ground, hit at the contact) are
// told apart as the reference's loop over its list does, first one wins (library.cu:243-251).  The first line skips
// the square root and the divides where no root can qualify: with t_min >= 0, a root beyond t_min needs
// h > 0 or the origin inside the sphere (ncc > 0); otherwise sqrt(disc) <= |h| and both roots are <= 0.
// orig = NULL: the device list is in the caller's order, and the first sphere met wins a tie by the strict bound alone.
// The sphere a bounce ray has just left is on every such ray's list (its origin is on the surface: "inside" to within the
// filter's slack), and for half of them ncc comes out a rounding error above zero, with the centre behind the ray
// (h <= 0).  Then only the far root (h + s) / a can lie beyond t_min, and it does not if
//   disc <= K^2 (1 - 2^-19),  K = t_min a - h >= 0:
// s = sqrt(disc) <= K (1 - 2^-22) even after its rounding, so h + s <= t_min a - 2^-22 K, rounded at most
// t_min a (1 - 2^-23) (|h + s| <= K, K >= t_min a), and the quotient rounds below t_min: the exact test would reject, and
// its square root and two divides are not needed (K, its square and the factor are rounded here: 4 x 2^-24 against the
// 2^-20 to spare).  With a = 0 the condition holds only for h = 0, where the roots are 0 / 0.
RT_DEV void accept_exact(float h, float ncc, float disc, float a, float t_min, int idx, const uint32_t* __restrict__ orig, Best& b) {
    bool maybe = disc >= 0.0f && (h > 0.0f || ncc > 0.0f);
#ifndef RT_EXP_NO_LEFT_TEST
    if (maybe && !(h > 0.0f)) {
        const float k = fmaf(t_min, a, -h);
        maybe = !(disc <= k * k * (1.0f - 0x1p-19f));
    }
#endif
    if (maybe) {
#ifdef RT_ABL_EXACT   // ablation (changes the image): what the exact test's IEEE square root and divides cost
        float s = __builtin_amdgcn_sqrtf(disc);
        const float ia = __builtin_amdgcn_rcpf(a);
        float root_a = (h - s) * ia;
#else
        float s = sqrtf(disc);
        float root_a = (h - s) / a;
#endif
        float root = root_a;
        bool ok = (t_min < root_a) && (root_a < b.t);
        // The far root is looked at only when the near one is not beyond t_min: otherwise the near one failed on the
        // upper bound, and the far one, (h + s) / a >= (h - s) / a (s >= 0, a > 0; addition and division round
        // monotonically), fails on it too -- its divide is skipped for most candidates.
        if (!(t_min < root_a)) {
#ifdef RT_ABL_EXACT
            float root_b = (h + s) * ia;
#else
            float root_b = (h + s) / a;
#endif
            root = root_b;
            ok = (t_min < root_b) && (root_b < b.t);
        }
        if (orig && root == b.t && b.idx >= 0 && t_min < root) ok = orig[idx] < orig[b.idx];   // a tie: the earlier sphere of the caller's list
        if (ok) { b.t = root; b.idx = idx; }
    }
}

// The three per-sphere terms, in the checker's op order:
//   oc = c - o; h = d.oc; ncc = r*r - oc.oc (accumulated from r*r down); disc = h*h + a*ncc
RT_DEV void sphere_terms(float cx, float cy, float cz, float rr, V3 o, V3 d, float a, float& h, float& ncc, float& disc) {
    float ocx = cx - o.x, ocy = cy - o.y, ocz = cz - o.z;
    h = fmaf(d.z, ocz, fmaf(d.y, ocy, d.x * ocx));
    ncc = fmaf(-ocz, ocz, fmaf(-ocy, ocy, fmaf(-ocx, ocx, rr)));
    disc = fmaf(h, h, a * ncc);
}

// LDS sphere tile of one workgroup:
//   geo[i] = {cx - Tx, cy - Ty, cz - Tz, m}   scan filter only (T, m: see scan_tile)
//   exa[i] = {cx, cy, cz, r*r}                exact test (LDS, or global memory in the queue kernels)
// box: two float4 per block of 32 spheres (scan_tile), or NULL; orig: the caller's index of every device sphere (the
// whole list's, whatever the tile)
// tbox: two float4, the box around the whole tile's spheres (the kernels that stream the list), or NULL
// The caller's index of every device sphere, behind the material tags -- where the launcher sorted the list (RtLaunch.sorted).
RT_DEV const uint32_t* orig_of(const RtLaunch& L, const uint32_t* __restrict__ mat) { return L.sorted ? mat + L.n_padded : nullptr; }

struct Tile { const float4* geo; const float4* exa; const float4* box; const uint32_t* orig; const float4* tbox; };

// Per-lane list of candidate spheres of the tile being scanned, kept in LDS as a column of a
// [cap][64] u16 array per wavefront (lane l owns column l, so pushes never conflict).
struct Cand { uint16_t* col; uint32_t n; uint32_t cap; };   // cap: slots per lane (wave-uniform)

RT_DEV void push(Cand& c, uint32_t idx) {
    c.col[c.n * 64u] = (uint16_t)idx;
    c.n++;
}

// Runs the exact test on every listed sphere, in list (= index) order.  Lanes with shorter lists idle
// for the extra rounds.
RT_DEV void drain(const Tile& tile, int base, Cand& c, V3 o, V3 d, float a, float t_min, Best& b, Stats& st) {
    st.wave(6);
#ifdef RT_EXP_DRAIN2
    for (int twice = 0; twice < 2; twice++)   // cost experiment: a second pass changes nothing (the bound is strict)
#endif
    for (uint32_t k = 0; k < c.n; k++) {                        // a lane leaves when its list is done
        st.wave(3);
        uint32_t i = c.col[k * 64u];
        float4 g = tile.exa[i];
        float h, ncc, disc;
        sphere_terms(g.x, g.y, g.z, g.w, o, d, a, h, ncc, disc);
        accept_exact(h, ncc, disc, a, t_min, base + (int)i, tile.orig, b);
    }
    c.n = 0u;
}

// The clear bits of one block's mask (bit j: sphere blk + (k-1-j)) become entries of the lanes' candidate
// lists, lowest sphere index first; full lists are drained on the spot.
// MAPPED with map != NULL (wave-uniform): the scanned entries are a pixel unit's list (unit_list) and map[] holds
// their sphere indices, ascending.
template <bool MAPPED>
RT_DEV void decode_block(const Tile& tile, int base, uint32_t blk, uint32_t k, uint32_t miss, const uint16_t* map,
                         V3 o, V3 d, float a, float t_min, Best& b, Cand& c, Stats& st) {
    uint32_t bits = ~miss & (k == 32u ? 0xffffffffu : ((1u << k) - 1u));
#ifdef RT_EXP_DECODE2
    {   // cost experiment: the decode loop once more into the same slots, then forgotten
        uint32_t bits2 = bits;
        const uint32_t n0 = c.n;
        while (__ballot(bits2 != 0u) != 0ull) {
            if (bits2 != 0u) {
                uint32_t pos = 31u - (uint32_t)__clz((int)bits2);
                push(c, blk + (k - 1u - pos));
                bits2 &= ~(1u << pos);
            }
            if (__ballot(c.n >= c.cap) != 0ull) c.n = n0;
        }
        c.n = n0;
    }
#endif
    while (bits != 0u) {                                            // a lane leaves when its bits are used up
        st.wave(4);
        st.add(2, 1u);
        uint32_t pos = 31u - (uint32_t)__clz((int)bits);            // highest bit = lowest sphere index first
        uint32_t idx = blk + (k - 1u - pos);
        if (MAPPED && map) idx = map[idx];
        push(c, idx);
        bits &= ~(1u << pos);
        // a full list: the lanes still decoding drain theirs together (the others' lists have room left)
        if (__ballot(c.n >= c.cap) != 0ull) { st.lap(10); drain(tile, base, c, o, d, a, t_min, b, st); st.lap(11); }
    }
    st.lap(10);
}

// 1/|d| for the scan filter, times 1 + 2^-19.  v_rsq_f32 is accurate to 1 ulp, which is also what a correctly
// rounded square root followed by a correctly rounded divide guarantee, so the derivation below holds as it
// stands; the filter is not part of the arithmetic contract (only the exact test decides), and the hardware
// instruction replaces some 25 operations per scan.  d.d = 0 or denormal gives +inf: every key is then +inf or a
// (positive) NaN, i.e. a candidate, and the exact test rejects.
RT_DEV float filter_inv_len(float a) { return __builtin_amdgcn_rsqf(a) * (1.0f + 0x1p-19f); }

// Scan `count` (a multiple of RT_SCAN_UNROLL) spheres staged in LDS; every lane of the wavefront reads
// the same float4 per sphere.  The scan only has to list, per lane, a SUPERSET of the spheres the exact
// test can accept; the exact test (sqrt, two divides, ~70 operations) then runs once per listed sphere
// with the lanes' lists side by side (drain), instead of once per sphere of the scan with one or two
// lanes active.  The closest hit does not depend on when a sphere is tested: the accepted root of a
// sphere is its first root beyond t_min, if below the bound, whatever the bound.
//
// Filter: the sphere can only be hit if  Q = r*r - |c-o|^2 + (u.(c-o))^2 >= 0  (u = d/|d|: squared radius
// minus squared distance from the centre to the ray's line).  Expanded around a scene-wide translation T
// (c' = c - T per sphere at upload, o' = o - T per ray) this is, per sphere and lane,
//   t = k0 + w.c'         (3 fma)   w = 2 o',  k0 = -|o'|^2 (1-2^-18)                   [per ray]
//   g = e + uf.c'         (3 fma)   uf = d rsq(d.d) (1+2^-19),  e = -uf.o'              [per ray]
//   q = g*g + t           (1 fma)
//   key = q + m           (1 add)   m = r*r(1+2^-17) - |c'|^2 (1-2^-18)                 [per sphere, upload]
// 8 VALU operations, +1 to record the sign of key: 9 per sphere, no branch, no compare, no subtraction
// of the origin from every centre.  The expansion cancels, so its rounding error grows with the
// magnitudes: below 35 * 2^-24 * (|c'|^2 + |o'|^2) in all (3 roundings of t at magnitude <= (|c'|+|o'|)^2,
// error of g times 2|g|, the roundings of q, key, k0, e and of the translated c', o').  The three
// inflations cover it: 2^-18 (|c'|^2 + |o'|^2) = 64 * 2^-24 * (...) for the expansion, and, for the
// difference between Q and the exact test's disc/a (whose own roundings scale with |c-o|^2 and r*r), the
// factors 1+2^-19 on uf and 1+2^-17 on r*r: with e = 2^-24 the exact test accepts only if
// h*h >= a(1-e)(oc.oc - rr - 3e(rr + oc.oc)), and uf.oc = s*h + err with |err| <= 8e|oc|, which gives
// Q_computed >= oc.oc (2^-18 - 40e) + rr (2^-17 - 2^-18 + 7e) > 0.  So disc >= 0 implies key >= 0.
// T (the component-wise median of the centres) keeps |c'| and |o'| of the order of the scene's extent
// instead of its distance from the world origin.  A NaN anywhere yields a candidate or a sphere the exact
// test rejects as well.  The sign bits of key for up to 32 consecutive spheres are shifted into one
// register per lane (v_alignbit: miss = miss << 1 | key >> 31); after each block the few clear bits
// become entries of the lane's candidate list.
// (Reading the filter data with scalar loads instead -- wave-uniform global addresses, the four values of a
// sphere as SGPR operands -- was built and measured 19 % slower than the LDS tile on the 485-sphere scene.)
//
// Scale.  The whole expression is evaluated times sigma^2, sigma a power of two chosen per scene at upload
// (RtLaunch.scan_scale): uf and e carry sigma, w and k0 sigma^2, and the uploaded m is m sigma^2.  Scaling by a power
// of two commutes with every rounding, so the sign of key is the same bit for bit.
//
// HALF = true (bounce rays of the camera-ray-batch kernels): spheres whose centre lies BEHIND the ray are dropped
// unless the origin is inside them.  g = uf.(c - o) is the (inflated) distance along the ray to the point nearest the
// centre; with g < 0 that point is behind the origin, and the exact test can then accept only if the origin is inside
// the sphere: its acceptance needs h > 0 or ncc > 0 (accept_exact), and h = d.oc > 0 with g < 0 means |u.oc| is
// below g's rounding error (8 * 2^-24 |oc|), so disc >= 0 needs |oc|^2 - r*r below 64 * 2^-48 |oc|^2 -- also "inside"
// to within what the inflations cover.  So for g < 0 the key is t + m alone (r*r - |c-o|^2, inflated), i.e.
// key = gc*gc + t + m with gc = max(g, 0).  The max costs nothing: sigma is chosen so that sigma |c - o| <= 1/2 for
// every origin on a sphere of the scene (the launcher keeps cameras too far away for that off these kernels), hence
// gc = clamp(g, 0, 1), the output modifier of the fma that produces g.  What goes: the mirror or the ground behind a
// ray (the sphere a ray has just left stays listed: its origin is "inside" to within the slack).  Candidates per ray
// 1.85 -> 1.58 on the 485-sphere book scene, 2.16 -> 1.80 in the mirror hall; exact-test rounds per scan 5.2 -> 3.4
// and 3.1 -> 2.5.
// (NaN: the clamp turns it into 0, key = t + m decides; the exact test rejects every sphere for a NaN ray anyway.)
// per-ray constants of the filter
struct RayTerms { V3 w; float k0; V3 uf; float e; };

// N spheres (8, 4 or 2): N 16-byte LDS reads, then 9 N operations; the sign bits are shifted into `miss`
template <bool HALF, uint32_t N>
RT_DEV void filter_step(const RayTerms& r, const float4* p, uint32_t& miss) {
    float4 s[N];
#pragma unroll
    for (uint32_t u = 0; u < N; u++) s[u] = p[u];
#pragma unroll
    for (uint32_t u = 0; u < N; u++) {
        const float t = fmaf(r.w.z, s[u].z, fmaf(r.w.y, s[u].y, fmaf(r.w.x, s[u].x, r.k0)));
        float g = fmaf(r.uf.z, s[u].z, fmaf(r.uf.y, s[u].y, fmaf(r.uf.x, s[u].x, r.e)));
        if (HALF) g = __builtin_amdgcn_fmed3f(g, 0.0f, 1.0f);      // folded into the fma's clamp output modifier
        const float key = fmaf(g, g, t) + s[u].w;
        miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(key), 31u);
    }
}

// A block of 32 spheres that few lanes reach (with 64 independent bounce rays most reached blocks are reached by one or
// two) is scanned the other way round.  Lane j holds sphere (j & 31) of the block -- one 16-byte LDS read per lane
// instead of 32 broadcast reads -- and the reaching rays (`reached`: their lanes) are taken two at a time, one per half
// of the wavefront, each lane fetching its ray's eight filter terms from that ray's lane: the same nine operations on
// the same numbers as filter_step, so the same keys bit for bit, for 32 spheres at once.  Some 15 instructions per
// reaching ray instead of 300 per block.  Called by the whole wavefront (every lane enabled); returns the lane's mask as filter_step builds it
// (all ones for a lane that does not reach the block).
template <bool HALF>
RT_DEV uint32_t sparse_block(const RayTerms& r, const float4* geo_blk, unsigned long long reached) {
    const uint32_t lane = threadIdx.x & 63u;
    const float4 sp = geo_blk[lane & 31u];
    uint32_t miss = 0xffffffffu;
    unsigned long long todo = reached;
    while (todo != 0ull) {
        // two rays per step: lanes 0..31 test the block's spheres for the first, lanes 32..63 for the second (the same
        // ray twice when only one is left); each lane fetches its ray's terms from that ray's lane (ds_bpermute:
        // the LDS crossbar, no memory access)
        const int l1 = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        int l2 = l1;
        if (todo != 0ull) { l2 = __ffsll((long long)todo) - 1; todo &= todo - 1ull; }
        const int src = (lane < 32u ? l1 : l2) << 2;
        const float wx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r.w.x)));
        const float wy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r.w.y)));
        const float wz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r.w.z)));
        const float k0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r.k0)));
        const float ux = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r.uf.x)));
        const float uy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r.uf.y)));
        const float uz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r.uf.z)));
        const float e0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(r.e)));
        const float t = fmaf(wz, sp.z, fmaf(wy, sp.y, fmaf(wx, sp.x, k0)));
        float g = fmaf(uz, sp.z, fmaf(uy, sp.y, fmaf(ux, sp.x, e0)));
        if (HALF) g = __builtin_amdgcn_fmed3f(g, 0.0f, 1.0f);
        const float key = fmaf(g, g, t) + sp.w;
        // bit j of a half of the ballot: sphere j of the block is a candidate for that half's ray (sign bit of its key
        // clear); the masks of decode_block count the other way: bit j set = sphere 31 - j cannot be hit
        const unsigned long long cand = __ballot(__float_as_int(key) >= 0);
        if (lane == (uint32_t)l1) miss = ~__builtin_bitreverse32((uint32_t)cand);
        if (lane == (uint32_t)l2) miss = ~__builtin_bitreverse32((uint32_t)(cand >> 32));
    }
    return miss;
}

// Slab test of a ray's half-line against a box {lo, hi} (coordinates relative to T), see scan_tile.
struct BoxTerms { V3 inv, n_lo, n_hi; };
RT_DEV BoxTerms box_terms(V3 op, V3 uf, float sigma) {
    BoxTerms bt;
    asm volatile("" : "+s"(sigma));                               // (made here, per ray, rather than kept through the render loop)
    const float lim = 0x1p40f * __builtin_amdgcn_rcpf(sigma);
    bt.inv = v3(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(uf.x), -lim, lim), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(uf.y), -lim, lim),
                __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(uf.z), -lim, lim));
    const float grow = __builtin_amdgcn_sqrtf(vdot(op, op)) * (0x1p-9f * (1.0f + 0x1p-8f));
    bt.n_lo = v3(-(op.x + grow) * bt.inv.x, -(op.y + grow) * bt.inv.y, -(op.z + grow) * bt.inv.z);   // lower planes move down by `grow`: plane - (o + grow)
    bt.n_hi = v3(-(op.x - grow) * bt.inv.x, -(op.y - grow) * bt.inv.y, -(op.z - grow) * bt.inv.z);
    return bt;
}
// tn, tf: where the half-line enters and leaves the box; it reaches the box unless tf < tn
RT_DEV void box_span(const BoxTerms& bt, float4 lo, float4 hi, float& tn, float& tf) {
    const float ax = fmaf(lo.x, bt.inv.x, bt.n_lo.x), bx = fmaf(hi.x, bt.inv.x, bt.n_hi.x);
    const float ay = fmaf(lo.y, bt.inv.y, bt.n_lo.y), by = fmaf(hi.y, bt.inv.y, bt.n_hi.y);
    const float az = fmaf(lo.z, bt.inv.z, bt.n_lo.z), bz = fmaf(hi.z, bt.inv.z, bt.n_hi.z);
    tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
}
// the filter's scaled unit direction of a ray (scan_tile)
RT_DEV V3 filter_dir(V3 d, float a, float sigma) {
    const float inv_len = filter_inv_len(a) * sigma;
    return v3(d.x * inv_len, d.y * inv_len, d.z * inv_len);
}

// the filter's per-ray constants (scan_tile); op = o - T
RT_DEV RayTerms ray_terms(V3 op, V3 d, float a, float sigma) {
    // (sigma, a launch constant in a scalar register, passes through an empty volatile asm: its square and twice its
    // square are then two instructions here instead of two vector registers held through the render loop)
    float sg = sigma;
    asm volatile("" : "+s"(sg));
    const float s2 = sg * sg;
    RayTerms r;
    r.uf = filter_dir(d, a, sigma);
    r.e = -vdot(r.uf, op);
    r.w = v3((2.0f * s2) * op.x, (2.0f * s2) * op.y, (2.0f * s2) * op.z);
    r.k0 = -(vdot(op, op) * (1.0f - 0x1p-18f)) * s2;
    return r;
}

// act (BOXES builds only): the lane has a ray.  Those builds call the scan with EVERY lane of the wavefront (a 64-wide
// instruction costs the same whatever the execution mask), because a block that few lanes reach is scanned the other
// way round -- lanes over spheres, see below -- and that needs the lanes whatever they hold; a lane without a ray
// reaches no box and lists nothing.
template <bool HALF, bool BOXES = false>
RT_DEV void scan_tile(const Tile& tile, uint32_t count, int base, V3 T, float sigma,
                      V3 o, V3 d, float a, float t_min, Best& b, Cand& c, Stats& st, bool act = true) {
    st.wave(1);
    const V3 op = vsub(o, T);
    const RayTerms r = ray_terms(op, d, a, sigma);
    // Block boxes (tile.box != NULL): the list is sorted in space and every block of 32 spheres has a box outside
    // which none of them can be hit; a block whose box no lane's half-line reaches is not scanned.  Slab test with
    // the box grown by 2^-9 |o - T| for this ray (see the launcher: the exact test's tolerance).  A plane's parameter is
    // (plane - o') / u_k = plane * inv_k - o' * inv_k: one fma per plane, the second product once per ray.  u = uf, the
    // filter's scaled unit direction (any positive multiple of d serves: the test compares parameters with each other and
    // with 0), and inv_k = 1 / u_k is clamped to +-2^40 / sigma -- as if no component of the unit direction were below
    // 2^-40: over the scene's extent that moves the ray by less than 2^-37 R, which the boxes' own margin (2^-19 R)
    // covers, and it keeps every product finite (1 / 0 = inf would give inf - inf for both planes of an axis the ray does
    // not move along).  The roundings of the two-product form are below 2^-22 (|o'| + |plane|) in the plane's position,
    // also within the margins (2^-9 |o'| (1 + 2^-8) here, 2^-19 R per sphere at upload).  minNum / maxNum drop what is
    // not a number (only a NaN ray produces one).
    // tile.tbox: the same test for the whole tile first (a streamed list: most tiles are out of every lane's reach).
    BoxTerms bt;
    bt.inv = bt.n_lo = bt.n_hi = v3(0.0f, 0.0f, 0.0f);
    if (BOXES && tile.box) {
        bt = box_terms(op, r.uf, sigma);
        if (tile.tbox) {
            float tn, tf;
            box_span(bt, tile.tbox[0], tile.tbox[1], tn, tf);
            if (__ballot(act && !(tf < tn)) == 0ull) return;           // (the lanes' candidate lists are empty between tiles)
        }
    }
    uint32_t blk = 0u;
    for (; blk + 32u <= count; blk += 32u) {                           // full blocks: one address, constant offsets
        bool reach = act;                                              // this lane's ray can hit a sphere of the block
        if (BOXES && tile.box) {
            const float4 lo = tile.box[blk >> 4], hi = tile.box[(blk >> 4) + 1u];
            float tn, tf;
            box_span(bt, lo, hi, tn, tf);
#ifdef RT_BOXCHECK   // diagnostics: a lane whose ray misses the box but whose exact test accepts a sphere of the block
            if (act && tf < tn) {
                for (uint32_t j = 0; j < 32u; j++) {
                    const float4 e = tile.exa[blk + j];
                    float hh, nn, dd;
                    sphere_terms(e.x, e.y, e.z, e.w, o, d, a, hh, nn, dd);
                    Best tmp; tmp.t = INFINITY; tmp.idx = -1;
                    accept_exact(hh, nn, dd, a, t_min, (int)(blk + j), tile.orig, tmp);
                    if (tmp.idx >= 0 && rt_dbg_ptr) {
                        float* w = reinterpret_cast<float*>(rt_dbg_ptr + 1);
                        w[0] = o.x; w[1] = o.y; w[2] = o.z; w[3] = d.x; w[4] = d.y; w[5] = d.z;
                        w[6] = (float)(blk + j); w[7] = tmp.t; w[8] = tn; w[9] = tf; w[10] = lo.x; w[11] = lo.y; w[12] = lo.z;
                        w[13] = hi.x; w[16] = hi.y; w[17] = hi.z; w[18] = 0.0f; w[19] = T.x; w[20] = T.y; w[21] = T.z;
                    }
                }
            }
#endif
            reach = act && !(tf < tn);                                 // (anything that is not a number reaches the box)
            const unsigned long long reached = __ballot(reach);
#ifdef RT_STATS_REACH
            { const uint32_t k = (uint32_t)__popcll(reached); st.reach(2); if (k) st.reach(k <= 2u ? 3 : k <= 4u ? 4 : k <= 8u ? 5 : k <= 16u ? 6 : 7); }
#endif
            if (reached == 0ull) continue;
            if ((uint32_t)__popcll(reached) <= RT_SPARSE_MAX) {
                const uint32_t miss = sparse_block<HALF>(r, tile.geo + blk, reached);
                st.lap(9);
                decode_block<false>(tile, base, blk, 32u, miss, nullptr, o, d, a, t_min, b, c, st);
                continue;
            }
        }
        uint32_t miss = 0u;                                            // bit j set: sphere blk + (31-j) cannot be hit
        // (BOXES builds: four spheres per step -- the eight float4 of a longer step in flight cost 16 registers that the
        // loop's other state needs, and most blocks go through sparse_block anyway)
        constexpr uint32_t STEP = BOXES ? RT_SCAN_UNROLL_BOXES : RT_SCAN_UNROLL;
#pragma unroll
        for (uint32_t i = 0; i < 32u; i += STEP) filter_step<HALF, STEP>(r, tile.geo + blk + i, miss);
        if (BOXES && !reach) miss = 0xffffffffu;                       // nothing of this block for a lane that does not reach it
        st.lap(9);
        decode_block<false>(tile, base, blk, 32u, miss, nullptr, o, d, a, t_min, b, c, st);
    }
    if (blk < count) {                                                 // the last 2 .. 30 spheres (count is even)
        const uint32_t k = count - blk;
        uint32_t miss = 0u, i = 0u;
        constexpr uint32_t STEP = BOXES ? RT_SCAN_UNROLL_BOXES : RT_SCAN_UNROLL;
        for (; i + STEP <= k; i += STEP) filter_step<HALF, STEP>(r, tile.geo + blk + i, miss);
        if (STEP > 4u && (k & 4u)) { filter_step<HALF, 4u>(r, tile.geo + blk + i, miss); i += 4u; }
        if (k & 2u) filter_step<HALF, 2u>(r, tile.geo + blk + i, miss);
        if (BOXES && !act) miss = 0xffffffffu;
        st.lap(9);
        decode_block<false>(tile, base, blk, k, miss, nullptr, o, d, a, t_min, b, c, st);
    }
    drain(tile, base, c, o, d, a, t_min, b, st);
    st.lap(11);
}

// scan_tile for one staged tile of a streamed list (whole blocks, plus a short last step behind them where the list
// ends), for the lanes whose ray reaches the tile's box (the caller calls only if some lane's does).  The blocks' boxes are tested the way sparse_block scans: where at most RT_SPARSE_TILE_MAX lanes reach
// the tile, lane j holds the box of block (j & 15) and the reaching rays are taken one at a time, their nine slab
// terms in scalar registers -- 30 instructions per reaching ray instead of 25 per block for all 64 lanes; then only
// the blocks somebody reaches are visited at all.
// tile.exa is the WHOLE list's exact-test data and the candidates are listed by their index in the list (base + index
// in the tile): the lists are not drained here but by the caller, once per path segment after the last tile (and by
// decode_block where one fills up) -- an exact-test round waits for global memory, and a wavefront has rays for eight
// tiles of a round but only two or three candidates per ray in all.
RT_DEV void scan_tile_stream(const Tile& tile, uint32_t count, int base, const RayTerms& r, const BoxTerms& bt, V3 o, V3 d, float a,
                             float t_min, Best& b, Cand& c, Stats& st, bool act) {
    st.wave(1);
    // (the lane's own answer for this tile once more, from the staged box: a bit per tile kept from the vote would cost
    // the round two registers)
    bool tile_reach = act;
    if (tile.tbox) {
        float tn, tf;
        box_span(bt, tile.tbox[0], tile.tbox[1], tn, tf);
        tile_reach = act && !(tf < tn);
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_full = count >> 5;                               // <= 16: RT_Q_STREAM_TILE / 32
    const unsigned long long tr = __ballot(tile_reach);
    uint32_t mine = 0u;                                               // bit j: this lane's ray reaches block j of the tile
    uint32_t anybody = 0u;                                            // ... some lane's does (wave-uniform)
    if (!tile.box) {                                                  // no boxes: every block, every lane with a ray
        mine = tile_reach ? (1u << n_full) - 1u : 0u;
        anybody = tr != 0ull ? (1u << n_full) - 1u : 0u;
    } else if ((uint32_t)__popcll(tr) <= RT_SPARSE_TILE_MAX) {
        const uint32_t jb = lane & 15u;
        const float4 lo = tile.box[2u * jb], hi = tile.box[2u * jb + 1u];
        unsigned long long todo = tr;
        while (todo != 0ull) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            BoxTerms q;
            q.inv = v3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt.inv.x), l)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt.inv.y), l)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt.inv.z), l)));
            q.n_lo = v3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt.n_lo.x), l)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt.n_lo.y), l)),
                        __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt.n_lo.z), l)));
            q.n_hi = v3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt.n_hi.x), l)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt.n_hi.y), l)),
                        __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bt.n_hi.z), l)));
            float tn, tf;
            box_span(q, lo, hi, tn, tf);
            const uint32_t m16 = (uint32_t)__ballot(jb < n_full && !(tf < tn)) & 0xffffu;   // (lanes 16..63 repeat lanes 0..15)
            anybody |= m16;
            if (lane == (uint32_t)l) mine = m16;
        }
    } else {
        for (uint32_t j = 0u; j < n_full; j++) {
            float tn, tf;
            box_span(bt, tile.box[2u * j], tile.box[2u * j + 1u], tn, tf);
            const bool reach = tile_reach && !(tf < tn);
            if (reach) mine |= 1u << j;
            if (__ballot(reach) != 0ull) anybody |= 1u << j;
        }
    }
#ifdef RT_BOXCHECK   // diagnostics: a lane whose ray is kept off a block but whose exact test accepts a sphere of it
    if (act)
        for (uint32_t j = 0u; j < n_full; j++)
            if (!((mine >> j) & 1u))
                for (uint32_t i = 0; i < 32u; i++) {
                    const float4 e = tile.exa[(uint32_t)base + 32u * j + i];
                    float hh, nn, dd;
                    sphere_terms(e.x, e.y, e.z, e.w, o, d, a, hh, nn, dd);
                    Best tmp; tmp.t = INFINITY; tmp.idx = -1;
                    accept_exact(hh, nn, dd, a, t_min, base + (int)(32u * j + i), tile.orig, tmp);
                    if (tmp.idx >= 0 && rt_dbg_ptr) {
                        float* w = reinterpret_cast<float*>(rt_dbg_ptr + 1);
                        w[0] = o.x; w[1] = o.y; w[2] = o.z; w[3] = d.x; w[4] = d.y; w[5] = d.z;
                        w[6] = (float)(base + (int)(32u * j + i)); w[7] = tmp.t;
                    }
                }
#endif
    while (anybody != 0u) {
        const uint32_t j = (uint32_t)__builtin_ctz(anybody);
        anybody &= anybody - 1u;
        const uint32_t blk = 32u * j;
        const bool reach = ((mine >> j) & 1u) != 0u;
        const unsigned long long reached = __ballot(reach);
        uint32_t miss;
        if ((uint32_t)__popcll(reached) <= RT_SPARSE_MAX) {
            miss = sparse_block<true>(r, tile.geo + blk, reached);
        } else {
            miss = 0u;
#pragma unroll
            for (uint32_t i = 0; i < 32u; i += RT_SCAN_UNROLL_BOXES) filter_step<true, RT_SCAN_UNROLL_BOXES>(r, tile.geo + blk + i, miss);
            if (!reach) miss = 0xffffffffu;
        }
        st.lap(9);
        decode_block<false>(tile, 0, (uint32_t)base + blk, 32u, miss, nullptr, o, d, a, t_min, b, c, st);
    }
    if (count & 31u) {                                                 // the list's last 2 .. 30 spheres: no box, every lane with a ray
        const uint32_t blk = 32u * n_full, k = count - blk;
        uint32_t miss = 0u, i = 0u;
        for (; i + RT_SCAN_UNROLL_BOXES <= k; i += RT_SCAN_UNROLL_BOXES) filter_step<true, RT_SCAN_UNROLL_BOXES>(r, tile.geo + blk + i, miss);
        if (RT_SCAN_UNROLL_BOXES > 4u && (k & 4u)) { filter_step<true, 4u>(r, tile.geo + blk + i, miss); i += 4u; }
        if (k & 2u) filter_step<true, 2u>(r, tile.geo + blk + i, miss);
        if (!act) miss = 0xffffffffu;
        st.lap(9);
        decode_block<false>(tile, 0, (uint32_t)base + blk, k, miss, nullptr, o, d, a, t_min, b, c, st);
    }
}

// The scan for camera rays.  They all leave from the camera centre o, and their directions are an affine function
// of the image coordinates: d = P0 + fx du + fy dv (P0 = pixel00 - o; make_ray).  With q = c - o the filter's
// condition  r*r - |q|^2 + (u.q)^2 >= 0  reads  |d.q| / |d| >= sqrt(|q|^2 - r*r),  and d.q = C + fx A + fy B with
// three numbers per sphere, A = du.q, B = dv.q, C = P0.q.  rt_camera_terms (below) computes them once per launch, in
// double precision, together with the right-hand side: per sphere {A, B, C, sq} in one 16-byte LDS read, and the
// scan is  dq = clamp(fma(fx, A, fma(fy, B, C)), 0, 1);  key = fma(dq, iLs, -sq)  and the sign bit: 4 operations per
// sphere and lane (the clamp is an output modifier), against 9 for a ray with an origin of its own.  The clamp's lower
// end drops spheres whose centre lies behind the ray (d.q < 0) unless the camera is inside them (sq = -inf below): with
// the camera outside, the exact test accepts only if h = d.q > 0 (accept_exact), and a finite sq > 0 means that the
// camera is outside by more than every rounding involved, s_abs included.  Its upper end is never reached: rt_camera_terms
// multiplies A, B, C and sq by rho, a power of two with rho * M < 1/2 for every sphere (M below bounds |dq|), which
// changes no sign.  Slack (so that the exact
// test's  disc >= 0  implies  key >= 0):  iLs = rsq(d.d)(1 + 2^-19);  sq = sqrt(|q|^2 (1-2^-18) - r*r (1+2^-17))
// (1 - 2^-20) - s_abs, rounded down, with the same two inflations as scan_tile for the difference between the
// filter's condition and the exact test's own roundings, and s_abs for this form's: 4 * 2^-24 M / F for the
// roundings of A, B, C and the two fma (M = |C| + (W+1)|A| + (H+1)|B| bounds every partial sum, F = the distance
// from o to the image plane <= |d|; fx and fy are the very numbers make_ray uses), 2 |q| eta / F for the difference
// in direction and in length between P0 + fx du + fy dv and make_ray's rounded d (eta: see rt_camera_terms), and
// 2^-22 |q| for the last fma.  A camera inside the sphere (or a NaN) gives sq = -inf: always a candidate; padding
// gives sq = +inf: never.  A camera ON a sphere, or so close to it that nothing is left of the bound above, but not
// inside it by the exact test's own single-precision ncc = r*r - |c - o|^2 <= 0 (the same number for every camera
// ray), can still only be hit by rays with h = d.q > 0: such a sphere gets the half-space test alone, C + delta in
// place of C and the smallest positive sq, delta bounding every rounding between the scan's dq and the exact test's h
// -- the mirror the reflection preset's camera sits on is then listed by no camera ray instead of by all of them.
template <uint32_t N>
RT_DEV void primary_step(float fx, float fy, float iLs, const float4* p, uint32_t& miss) {
    float4 s[N];
#pragma unroll
    for (uint32_t u = 0; u < N; u++) s[u] = p[u];
#pragma unroll
    for (uint32_t u = 0; u < N; u++) {
        // clamp(fx A + fy B + C, 0, 1): the median of three folds into the fma's clamp output modifier
        const float dq = __builtin_amdgcn_fmed3f(fmaf(fx, s[u].x, fmaf(fy, s[u].y, s[u].z)), 0.0f, 1.0f);
        const float key = fmaf(dq, iLs, -s[u].w);
        miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(key), 31u);
    }
}

// geo0: `count` entries {A, B, C, sq} in LDS -- the tile's spheres (map = NULL), or a pixel unit's list with the
// spheres' indices in map[] (unit_list below).
RT_DEV void scan_primary(const float4* geo0, const uint16_t* map, const Tile& tile, uint32_t count, int base,
                         float fx, float fy, V3 o, V3 d, float a, float t_min, Best& b, Cand& c, Stats& st) {
    st.wave(0);
    const float iLs = filter_inv_len(a);
    // (four spheres per step: a unit's list has a handful of entries, and the eight float4 of a longer step in flight
    // cost the caller -- which holds its lanes' bounce paths meanwhile -- 16 registers)
    constexpr uint32_t STEP = RT_SCAN_UNROLL_PRIMARY;
    uint32_t blk = 0u;
    for (; blk + 32u <= count; blk += 32u) {
        uint32_t miss = 0u;
#pragma unroll
        for (uint32_t i = 0; i < 32u; i += STEP) primary_step<STEP>(fx, fy, iLs, geo0 + blk + i, miss);
        st.lap(15);
        decode_block<true>(tile, base, blk, 32u, miss, map, o, d, a, t_min, b, c, st);
    }
    if (blk < count) {
        const uint32_t k = count - blk;
        uint32_t miss = 0u, i = 0u;
        for (; i + STEP <= k; i += STEP) primary_step<STEP>(fx, fy, iLs, geo0 + blk + i, miss);
        if (STEP > 4u && (k & 4u)) { primary_step<4u>(fx, fy, iLs, geo0 + blk + i, miss); i += 4u; }
        if (k & 2u) primary_step<2u>(fx, fy, iLs, geo0 + blk + i, miss);
        st.lap(15);
        decode_block<true>(tile, base, blk, k, miss, map, o, d, a, t_min, b, c, st);
    }
    drain(tile, base, c, o, d, a, t_min, b, st);
    st.lap(11);
}

// The camera rays of one pixel unit see few spheres, and all of its rays nearly the same ones (16 pixels x some
// hundred samples, each ray within half a pixel of its pixel's centre).  So the scan over the whole list is done once
// per UNIT instead of once per 64 rays: unit_list tests every sphere against the unit's bundle of rays, and the few
// that can be hit by any of them -- about two per unit on the 485-sphere book scene at 4K, three on the 16,385-sphere
// grid -- are copied, camera-ray terms and index, to a short list in LDS, lowest index first.  The unit's camera-ray
// batches then run scan_primary over that list instead of over the scene.  Nothing changes in what the exact test
// sees, except that it sees it sooner.
//   The test is geometric: distance from the sphere's centre to the line through the centre of the unit's footprint
//   against the sphere's radius plus the bundle's half-width at that depth.  (The single-precision cone test of
//   scan_primary cannot resolve a small far sphere from its neighbours: its slack for the rounding of the camera's
//   directions is a slack on a cosine near 1.)  Per unit, in double precision from the camera's single-precision
//   fields: u, the unit vector through the centre of the footprint (fx in [x0 - 1/2, x0 + w], fy in [y0 - 1/2, y0 + h]:
//   pixel + jitter in [-1/2, 1/2), or the grid sampler's [0, 1)), o' = o - T and
//     s = reach / |d_centre|       reach (RtLaunch.unit_reach) >= |d - d_centre| for every ray of the unit as make_ray
//                                  rounds it: (w + 1/2)/2 |du| + (h + 1/2)/2 |dv| + eta, eta bounding make_ray's four
//                                  roundings per component; so the sine of the angle between a ray and the central
//                                  line is at most s
//   all three rounded to single precision (u is then a unit vector to 2^-23).  Per sphere, single precision, with
//   q = c' - o' (c' = c - T as stored for the scan filter):
//     dist = |q x u|               distance from the centre to the central line, as a cross product: no cancellation
//                                  beyond 3 * 2^-24 |q| per component
//     r_eff^2 = r*r (1 + 2^-17) + 2^-18 |q|^2    the exact test accepts a ray only if the centre is within r_eff of the
//                                  ray's line (the inflations of scan_tile's derivation: its own roundings)
//   The centre's angle to a ray's line is at least its angle to the central line minus the angle between the two, and
//   sines of angles up to a right angle add at most: dist(ray) >= dist - |q| s.  A sphere is listed unless
//     dist > bound   or   u.q < -bound,     bound = r_eff + |q| (s + 2^-18) + 2^-20 (|o'| + |c'|)  (1-norms)
//   (the second: the centre lies behind every ray by more than r_eff, camera outside -- the exact test then has
//   h = d.q <= 0 and ncc <= 0 and rejects).  2^-18 |q| covers the roundings of the cross product, of u and of the exact
//   test's h; 2^-20 (|o'| + |c'|) the roundings of c', o' and q.  Hardware square roots (1 ulp) are rounded up by
//   2^-20.  Anything that is not a number lists the sphere.
// No list (RT_UL_NONE: the batches scan the scene as before) if reach >= |d_centre| -- a degenerate camera --, or if
// more spheres pass than the list holds.
#define RT_UL_NONE 0xffffffffu

RT_DEV uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct UnitList { float4* terms; uint16_t* idx; };             // RtLaunch.ulist_cap entries each, per wavefront or pool slot

// geom: {c - T, .} and exact: {., r*r} per sphere (global memory); terms: the launch's camera-ray terms (LDS copy, or
// global memory where the sphere list is streamed), copied for the listed spheres.
// BLOCKS (the streaming kernels, whose lists run to tens of thousands of spheres): the boxes of the blocks of 32 are
// tested first, 64 blocks per step, and only the spheres of the blocks that pass are looked at, a block per step.  A
// block's box holds its spheres whole, so does the ball round the box (centre cb, radius rb = half the diagonal), and
// the sphere test above carries over with the ball in the sphere's place and its slacks summed instead of squared:
// a ray the exact test can accept passes within r_eff <= r (1 + 2^-17) + 2^-9 |q| of its sphere's centre, |q| <= |qb| +
// rb, so within rb (1 + 2^-8) + 2^-9 |qb| of cb; a block is skipped if
//     dist_b > bound_b   or   u.qb < -bound_b,     bound_b = rb (1 + 2^-7) + (|qb| + rb) (s + 2^-8) + 2^-19 (|o'| + |cb|)
// (the second: u.q <= u.qb + |c' - cb| < -bound_b + rb - r, which is below the sphere's own -bound).  The host stores
// the ball per block, {cb, rb (1 + 2^-7)}, the radius rounded up and with the rounding of cb added; an empty block or
// one with a box that is not finite has rb = +inf and passes; its padding spheres fail the sphere test.
struct UnitBundle { float ux, uy, uz, opx, opy, opz, s; };     // the unit's central direction, o - T, and the bundle's sine

// The scan of unit_list.
template <bool BLOCKS>
RT_DEV uint32_t unit_list_scan(const float4* __restrict__ geom, const float4* __restrict__ exact, const float4* terms, float4* ul_terms,
                               uint16_t* ul_idx, UnitBundle ub, uint32_t n_padded, uint32_t scan_blocks, uint32_t ulist_cap, uint32_t lane) {
    const float ux = ub.ux, uy = ub.uy, uz = ub.uz, opx = ub.opx, opy = ub.opy, opz = ub.opz, s = ub.s;
    const float o_abs = 0x1p-20f * (fabsf(opx) + fabsf(opy) + fabsf(opz));
    uint32_t n = 0u;
    auto sphere_passes = [&](uint32_t i) {
        const float4 g = geom[i];
        const float rr = exact[i].w;
        const float qx = g.x - opx, qy = g.y - opy, qz = g.z - opz;
        const float qq = (fmaf(qz, qz, fmaf(qy, qy, qx * qx))) * (1.0f + 0x1p-20f);
        const float t = fmaf(uz, qz, fmaf(uy, qy, ux * qx));
        const float kx = fmaf(qy, uz, -(qz * uy)), ky = fmaf(qz, ux, -(qx * uz)), kz = fmaf(qx, uy, -(qy * ux));
        const float dist2 = fmaf(kz, kz, fmaf(ky, ky, kx * kx));
        const float r_eff = __builtin_amdgcn_sqrtf(fmaf(rr, 1.0f + 0x1p-17f, 0x1p-18f * qq) * (1.0f + 0x1p-20f)) * (1.0f + 0x1p-20f);
        const float q_len = __builtin_amdgcn_sqrtf(qq) * (1.0f + 0x1p-20f);
        const float bound = (fmaf(q_len, s, r_eff) + fmaf(0x1p-20f, fabsf(g.x) + fabsf(g.y) + fabsf(g.z), o_abs)) * (1.0f + 0x1p-20f);
        return !(dist2 > bound * bound) && !(t < -bound) && rr != -INFINITY;   // (r*r = -inf: padding)
    };
    // appends the passing lanes' spheres in lane order; false: the list is full
    auto append = [&](bool pass, uint32_t i) {
        const unsigned long long m = __ballot(pass);
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (n + cnt > ulist_cap) return false;
        if (pass) {
            const uint32_t r = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            ul_terms[r] = terms[i];
            ul_idx[r] = (uint16_t)i;
        }
        n += cnt;
        return true;
    };
    if (BLOCKS && scan_blocks != 0u) {
        // per block {cb, rb (1 + 2^-7) rounded up}, behind the blocks' and the tiles' boxes (rt_upload_scene)
        const float4* ball = geom + n_padded + 2u * scan_blocks + 2u * ((n_padded + RT_Q_STREAM_TILE - 1u) / RT_Q_STREAM_TILE);
        for (uint32_t first = 0u; first < scan_blocks; first += 64u) {
            const uint32_t b = first + lane;
            bool reach = false;
            if (b < scan_blocks) {
                const float4 g = ball[b];
                const float qx = g.x - opx, qy = g.y - opy, qz = g.z - opz;
                const float q_len = __builtin_amdgcn_sqrtf(fmaf(qz, qz, fmaf(qy, qy, qx * qx))) * (1.0f + 0x1p-18f);
                const float t = fmaf(uz, qz, fmaf(uy, qy, ux * qx));
                const float kx = fmaf(qy, uz, -(qz * uy)), ky = fmaf(qz, ux, -(qx * uz)), kz = fmaf(qx, uy, -(qy * ux));
                const float dist2 = fmaf(kz, kz, fmaf(ky, ky, kx * kx));
                const float bound = (fmaf(q_len + g.w, s + 0x1p-8f, g.w) + fmaf(0x1p-19f, fabsf(g.x) + fabsf(g.y) + fabsf(g.z), 2.0f * o_abs)) * (1.0f + 0x1p-18f);
                reach = !(dist2 > bound * bound) && !(t < -bound);
            }
            unsigned long long todo = __ballot(reach);
            while (todo) {                                        // wave-uniform: a block per step, lowest first
                const uint32_t i = (first + (uint32_t)__builtin_ctzll(todo)) * 32u + lane;
                todo &= todo - 1ull;
                const bool pass = lane < 32u && i < n_padded && sphere_passes(i);
                if (!append(pass, i)) return RT_UL_NONE;
            }
        }
    } else {
        for (uint32_t first = 0u; first < n_padded; first += 64u) {
            const uint32_t i = first + lane;
            const bool pass = i < n_padded && sphere_passes(i);
            if (!append(pass, i)) return RT_UL_NONE;
        }
    }
    if (n & 1u) {                                                 // even, as the scans' last step wants it (the capacity is even)
        if (lane == 0u) { ul_terms[n] = make_float4(0.0f, 0.0f, 0.0f, INFINITY); ul_idx[n] = 0u; }
        n++;
    }
    return n;
}

template <bool BLOCKS = false>
RT_DEV uint32_t unit_list(const float4* __restrict__ geom, const float4* __restrict__ exact, const float4* terms, const RtCamera& cam,
                          const RtLaunch& L, uint32_t x0, uint32_t y0, const UnitList& ul, uint32_t lane) {
    float ux, uy, uz, opx, opy, opz, s;
    {
        // (the launch's constants pass through an empty volatile asm: otherwise everything below that does not depend on
        // the unit -- the conversions to double, half the unit's size, pixel00 - o -- is computed once, before the render
        // loop, and a dozen doubles sit in registers and in every lane's scratch through the loop)
        auto fresh = [](float v) { asm volatile("" : "+s"(v)); return (double)v; };
        uint32_t unit_w = L.unit_w, unit_h = L.unit_h;
        asm volatile("" : "+s"(unit_w), "+s"(unit_h));
        const double fxc = (double)x0 + 0.5 * ((double)unit_w - 0.5), fyc = (double)y0 + 0.5 * ((double)unit_h - 0.5);
        const double ox = fresh(cam.center[0]), oy = fresh(cam.center[1]), oz = fresh(cam.center[2]);
        const double dx = (fresh(cam.pixel00_loc[0]) - ox) + (fxc * fresh(cam.pixel_delta_u[0]) + fyc * fresh(cam.pixel_delta_v[0]));
        const double dy = (fresh(cam.pixel00_loc[1]) - oy) + (fxc * fresh(cam.pixel_delta_u[1]) + fyc * fresh(cam.pixel_delta_v[1]));
        const double dz = (fresh(cam.pixel00_loc[2]) - oz) + (fxc * fresh(cam.pixel_delta_u[2]) + fyc * fresh(cam.pixel_delta_v[2]));
        const double len2 = dx * dx + dy * dy + dz * dz;
        const float len_lo = __builtin_amdgcn_sqrtf((float)len2) * (1.0f - 0x1p-20f);
        if (!(len_lo > L.unit_reach)) return RT_UL_NONE;          // wave-uniform: every operand is
        s = L.unit_reach / len_lo * (1.0f + 0x1p-20f) + 0x1p-18f;
        const double inv = 1.0 / sqrt(len2);
        ux = (float)(dx * inv); uy = (float)(dy * inv); uz = (float)(dz * inv);
        opx = (float)(ox - fresh(L.translate[0])); opy = (float)(oy - fresh(L.translate[1])); opz = (float)(oz - fresh(L.translate[2]));
    }
    const UnitBundle ub = {ux, uy, uz, opx, opy, opz, s};
    return unit_list_scan<BLOCKS>(geom, exact, terms, ul.terms, ul.idx, ub, L.n_padded, L.scan_blocks, L.ulist_cap, lane);
}

// ---------------------------------------------------------------- shading
struct Scene {
    const float4* __restrict__ geom;    // filter data {c - T, m} per sphere, n_padded entries
    const float4* __restrict__ exact_;  // {c, r*r} per sphere, n_padded entries
    const float4* __restrict__ shade_;  // per sphere two float4: {cx, cy, cz, radius}, {albedo r, g, b, fuzz | ior}; after the
                                        // 2 * n_padded of those, one float4 per sphere with the glass constants (see scatter)
    const uint32_t* __restrict__ mat_;  // material tag per sphere
    const float4* __restrict__ camterms_;   // the launch's camera-ray terms per sphere (rt_camera_terms)
    RT_DEV const float4* exact() const { return exact_; }
    RT_DEV const float4* shade() const { return shade_; }
    RT_DEV const uint32_t* mat() const { return mat_; }
    RT_DEV const float4* camterms() const { return camterms_; }
};

// The same arrays found from the first one: rt_upload_scene puts a scene's arrays into one allocation -- filter data,
// the blocks' and tiles' boxes and the blocks' balls, then exact-test data, shading data, camera-ray terms, material
// tags -- so a kernel can hold ONE address and the sphere count where Scene holds five addresses.  The streaming
// kernels do: ten scalar registers they do not have (the values were kept in vector registers, and those in scratch).
// The count passes through an empty volatile asm at every use, so the sums are made where they are needed (a few
// scalar instructions) instead of once and kept.
struct SceneAt {
    const float4* __restrict__ geom;
    uint32_t n_padded;
    RT_DEV uint32_t n() const { uint32_t v = n_padded; asm volatile("" : "+s"(v)); return v; }
    RT_DEV static uint32_t head(uint32_t v) { return v + 3u * ((v + 31u) >> 5) + 2u * ((v + RT_Q_STREAM_TILE - 1u) / RT_Q_STREAM_TILE); }
    RT_DEV const float4* exact() const { const uint32_t v = n(); return geom + head(v); }
    RT_DEV const float4* shade() const { const uint32_t v = n(); return geom + (head(v) + v); }
    RT_DEV const float4* camterms() const { const uint32_t v = n(); return geom + (head(v) + 4u * v); }
    RT_DEV const uint32_t* mat() const { const uint32_t v = n(); return (const uint32_t*)(geom + (head(v) + 5u * v)); }
};

struct HitInfo { V3 p, n; bool front; };

// Hit record of the closest sphere: point, normal facing the ray (library.cu:220-225, lib.rs:118-122).
RT_DEV HitInfo finish_hit(float4 cr, float t, V3 o, V3 d, uint32_t flags) {
    HitInfo h;
    V3 c = v3(cr.x, cr.y, cr.z);
    const float radius = cr.w;
    h.p = v3(fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z));
    V3 pc = vsub(h.p, c);
#ifdef RT_ABL_HITDIV  // ablation (changes the image): what the hit record's three IEEE divides cost
    const float ir = __builtin_amdgcn_rcpf(radius);
    V3 n = v3(pc.x * ir, pc.y * ir, pc.z * ir);
#else
    V3 n = v3(pc.x / radius, pc.y / radius, pc.z / radius);
#endif
    float nd = vdot(n, d);
    bool flip = (flags & RT_FLAG_FLIP_SIGNBIT) ? !(__float_as_int(nd) < 0)   // lib.rs:49-58
                                               : (nd > 0.0f);                // floatN_helper.cu:129-136
    if (flip) n = v3(-n.x, -n.y, -n.z);
    h.n = n;
    h.front = !flip;
    return h;
}

// (1-a)*white + a*(0.5,0.7,1.0), a = blend*(unit(d).y + 1)  (ray.cu:43-45, lib.rs:203-205)
RT_DEV V3 sky(V3 d, float blend) {
    V3 u = vunit(d);
    float a = blend * (u.y + 1.0f);
    float om = 1.0f - a;
    return v3(fmaf(a, 0.5f, om), fmaf(a, 0.7f, om), fmaf(a, 1.0f, om));
}

// Material::scatter (library.cu:131-149,175-183) plus the book's dielectric.  The segment's stream
// (seed, pixel, sample, k) is started only by a material that draws from it: Lambertian and metal with
// fuzz > 0 draw the unit-vector tries, glass draws one uniform (reflect or refract), a mirror (fuzz 0) draws
// nothing -- a wavefront whose lanes all hit mirrors does no integer work here at all.
RT_DEV V3 scatter(uint32_t mat, float param, const float4* __restrict__ glass_terms, V3 d, const HitInfo& h, uint64_t seed,
                  uint32_t pixel, uint32_t s, uint32_t k, uint32_t flags, Stats& st) {
    const bool band = (flags & RT_FLAG_BAND_ALBEDO) != 0u;
    const bool glass = !band && mat == RT_MAT_DIELECTRIC;
    const bool wants_unit = band || mat == RT_MAT_DIFFUSE || (mat == RT_MAT_REFLECT && param != 0.0f);
    float u_choice = 0.0f;
    V3 ru = v3(0.0f, 0.0f, 0.0f);
    if (glass || wants_unit) {
        Rng g = stream_init(seed, pixel, s, k);
        if (glass) u_choice = rand01(g);
#ifdef RT_EXP_UNIT2
        if (wants_unit) {   // cost experiment: one more unit vector from a copy of the stream, kept alive and unused
            Rng g2 = g;
            g2.state ^= 0x9E3779B97F4A7C15ULL;
            V3 r2 = random_unit_vector(g2, st);
            asm volatile("" :: "v"(r2.x), "v"(r2.y), "v"(r2.z));
        }
#endif
        if (wants_unit) ru = random_unit_vector(g, st);
    }
    V3 n = h.n;
    V3 out;
    if ((flags & RT_FLAG_BAND_ALBEDO) || mat == RT_MAT_DIFFUSE) {
        out = vadd(n, ru);                                       // library.cu:132, lib.rs:194
        if (!(flags & RT_FLAG_BAND_ALBEDO)) {
            const float s = 1e-8f;                               // floatN_helper.cu:116-119
            if (fabsf(out.x) < s && fabsf(out.y) < s && fabsf(out.z) < s) out = n;
        }
    } else if (mat == RT_MAT_REFLECT) {
        float k = 2.0f * vdot(d, n);                             // floatN_helper.cu:124-127
        V3 refl = v3(fmaf(-n.x, k, d.x), fmaf(-n.y, k, d.y), fmaf(-n.z, k, d.z));
        V3 ur = vunit(refl);                                     // library.cu:146-147
        out = ur;                                                // fuzz 0: fuzz * random adds nothing
        if (wants_unit) out = v3(fmaf(ru.x, param, ur.x), fmaf(ru.y, param, ur.y), fmaf(ru.z, param, ur.z));
    } else {
        // Dielectric ("Ray Tracing in One Weekend" section 11): Snell refraction, Schlick reflectance.
        // the sphere's constants, computed once at upload with the same single-precision operations:
        // {1/ior, r0 entering, r0 leaving, -}, r0 = ((1 - ri) / (1 + ri))^2 for ri = 1/ior and ri = ior
        const float4 gt = *glass_terms;
        float ri = h.front ? gt.x : param;
        V3 ud = vunit(d);
        float udn = vdot(ud, n);
        float cos_t = fminf(-udn, 1.0f);
        float sin_t = sqrtf(fmaf(-cos_t, cos_t, 1.0f));
        bool cannot = ri * sin_t > 1.0f;
        float r0 = h.front ? gt.y : gt.z;
        float x = 1.0f - cos_t;
        float x2 = x * x;
        float x5 = x2 * x2 * x;
        float prob = fmaf(1.0f - r0, x5, r0);
        if (cannot || prob > u_choice) {
            float k = 2.0f * udn;
            out = v3(fmaf(-n.x, k, ud.x), fmaf(-n.y, k, ud.y), fmaf(-n.z, k, ud.z));
        } else {
            V3 perp = v3(ri * fmaf(cos_t, n.x, ud.x), ri * fmaf(cos_t, n.y, ud.y), ri * fmaf(cos_t, n.z, ud.z));
            float par = -sqrtf(fabsf(1.0f - vdot(perp, perp)));
            out = v3(fmaf(par, n.x, perp.x), fmaf(par, n.y, perp.y), fmaf(par, n.z, perp.z));
        }
    }
    return out;
}

// Primary ray of pixel (i,j) with sub-pixel offset (si,sj): ray.cu:62-77 / lib.rs:215-217.
RT_DEV void ray_through(const RtCamera& cam, float fx, float fy, V3& o, V3& d) {
    V3 pc;
    pc.x = cam.pixel00_loc[0] + fmaf(cam.pixel_delta_u[0], fx, cam.pixel_delta_v[0] * fy);
    pc.y = cam.pixel00_loc[1] + fmaf(cam.pixel_delta_u[1], fx, cam.pixel_delta_v[1] * fy);
    pc.z = cam.pixel00_loc[2] + fmaf(cam.pixel_delta_u[2], fx, cam.pixel_delta_v[2] * fy);
    o = ld3(cam.center);
    d = vsub(pc, o);
}

RT_DEV void make_ray(const RtCamera& cam, uint32_t i, uint32_t j, float si, float sj, V3& o, V3& d, float& fx, float& fy) {
    fx = (float)i + si;
    fy = (float)j + sj;
    ray_through(cam, fx, fy, o, d);
}

RT_DEV void sample_ray(const RtCamera& cam, const RtLaunch& L, uint64_t seed, uint32_t x, uint32_t y,
                       uint32_t pixel, uint32_t s, V3& o, V3& d, float& fx, float& fy) {
    float si, sj;
    if (L.flags & RT_FLAG_SAMPLER_GRID) {                        // ray.cu:59-60
        uint32_t n2 = L.sample2_per_pixel;
        si = (float)(s % n2) / (float)n2;
        sj = (float)(s / n2) / (float)n2;
    } else {                                                     // lib.rs:209-214
#ifdef RT_ABL_JITTER  // ablation (changes the image): what the jitter's stream start and two outputs cost
        si = (float)(s & 15u) * 0.0625f - 0.5f;
        sj = (float)((s >> 4) & 15u) * 0.0625f - 0.5f;
#else
        Rng g = stream_init(seed, pixel, s, 0u);
        si = rand01(g) - 0.5f;
        sj = rand01(g) - 0.5f;
#endif
    }
    make_ray(cam, x, y, si, sj, o, d, fx, fy);
}

RT_DEV uint32_t quant(float lin, uint32_t flags) {
    float v = sqrtf(lin) * 255.0f;                               // gamma 2: lib.rs:236, ray.cu:129
    if (flags & RT_FLAG_QUANT_ROUND) {                           // (x.sqrt()*255.0).round() as u8
        if (!(v == v)) return 0u;
        float r = roundf(v);
        if (r <= 0.0f) return 0u;
        if (r >= 255.0f) return 255u;
        return (uint32_t)r;
    }
    if (!(v == v)) return 255u;                                  // ray.cu:81-89: fminf drops a NaN
    float c = v < 255.0f ? v : 255.0f;
    c = c > 0.0f ? c : 0.0f;
    return (uint32_t)c;
}

// Wave tile -> pixel.  A wavefront owns an 8x8 pixel tile; consecutive tiles of a workgroup sit side
// by side in a row.  Returns false for lanes outside the image / row set.
RT_DEV bool map_pixel(const RtLaunch& L, uint32_t wave_tile, uint32_t lane, uint32_t& x, uint32_t& ly, uint32_t& y) {
    uint32_t tx = wave_tile % L.tiles_x, ty = wave_tile / L.tiles_x;
    x = tx * 8u + (lane & 7u);
    ly = ty * 8u + (lane >> 3u);
    uint32_t k = ly / L.tile_rows;
    y = L.row_base + k * L.tile_stride_rows + (ly - k * L.tile_rows);
    return wave_tile < L.n_wave_tiles && x < L.width && ly < L.rows_local && y < L.height;
}

// Dynamic LDS: [geo: S float4][exa: S float4][candidate lists: waves x RT_CAND_CAP x 64 u16]
// [accumulators: waves x 2 slots x P x 3 u64 (r, g, b of a pixel side by side), P = pixels per unit], S = min(n_padded, RT_TILE_SPHERES)
extern __shared__ float4 rt_lds[];

RT_DEV uint32_t tile_capacity(const RtLaunch& L) { return L.n_padded < RT_TILE_SPHERES ? L.n_padded : RT_TILE_SPHERES; }

RT_DEV Tile stage_tile(const RtLaunch& L, const float4* __restrict__ geom, const float4* __restrict__ exact,
                       const uint32_t* __restrict__ orig, uint32_t first, uint32_t count) {
    float4* geo = rt_lds;
    float4* exa = rt_lds + tile_capacity(L);
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) { geo[i] = geom[first + i]; exa[i] = exact[first + i]; }
    Tile t = {geo, exa, nullptr, orig};
    return t;
}

RT_DEV unsigned long long* accum_base(const RtLaunch& L) {
    uint16_t* cand = reinterpret_cast<uint16_t*>(rt_lds + 2u * tile_capacity(L));
    return reinterpret_cast<unsigned long long*>(cand + (blockDim.x >> 6) * (RT_CAND_CAP * 64u));
}

RT_DEV Cand cand_list(const RtLaunch& L) {
    uint16_t* base = reinterpret_cast<uint16_t*>(rt_lds + 2u * tile_capacity(L));
    Cand c;
    c.col = base + (threadIdx.x >> 6) * (RT_CAND_CAP * 64u) + (threadIdx.x & 63u);
    c.n = 0u;
    c.cap = RT_CAND_CAP;
    return c;
}

// ---------------------------------------------------------------- render
// Run sum -> 2^-32 fixed point (floor part + truncated fraction); values that are not finite and below
// 2^30 in magnitude count as 0.  See RtParams.sample_chunk.
RT_DEV long long to_fixed(float x) {
    if (!(fabsf(x) < 1073741824.0f)) return 0ll;
    float hi = floorf(x);
    float frac = x - hi;                                         // in [0,1], exact unless x is a tiny negative number:
    if (frac >= 1.0f) { hi += 1.0f; frac = 0.0f; }               // then hi = -1 and x + 1 rounds to 1
    uint32_t lo = (uint32_t)(frac * 4294967296.0f);              // truncation
    return (long long)(int)hi * 4294967296ll + (long long)lo;
}
// (spp, wave-uniform, passes through an empty volatile asm: its conversion to double is then made here, at a unit's
// write-out, instead of once ahead of the render loop and kept in two vector registers through it)
RT_DEV float fixed_mean(long long sum, uint32_t spp) {
    asm volatile("" : "+s"(spp));
    return (float)((double)sum * 0x1p-32 / (double)spp);
}

// Work distribution.  The frame (or this launch's rows of it) is cut into units of unit_w x unit_h pixels
// (8x8, 8x1, ...); a unit's samples are cut into work items of `chunk` consecutive samples of one
// pixel (item i: pixel i % P, run i / P).  The grid is sized to fill the GPU once; every wavefront takes
// units from a launch-wide counter until none are left.  A lane traces the paths of its item one after
// the other, adding their colours in fp32; when the run is complete it adds the run sum to the pixel's
// accumulator in LDS and takes the next item.  A wavefront keeps two units open: when the items of one
// are all handed out it starts on the next while the last, longest paths of the first finish, so the
// lanes stay busy whatever the length of the individual paths -- without this, a lane that owns a pixel
// of sky (one segment per sample) idles while its neighbour finishes a pixel of glass (tens of segments
// per sample).  A unit's pixels are written when its last item completes.
//   n_chunks == 1 (RtParams.sample_chunk = 0): one item per pixel, 8x8 units; the sum is the reference's
//     fp32 fold in sample order (ray.cu:123-129, lib.rs:228-231), kept by one lane per pixel.
//   n_chunks  > 1: run sums are accumulated exactly (2^-32 fixed point, 64-bit LDS atomics), so the result
//     does not depend on which lane traced which run, nor on the unit size.
// MULTI = false: the whole (padded) sphere list fits the LDS tile; staged once, after which every
//                wavefront runs on its own.
// MULTI = true : the list is streamed tile by tile; the workgroup moves in lock step, one path
//                segment per outer iteration.
struct Item {
    uint32_t x, y, pixel;       // image coordinates, global pixel index (RNG key)
    uint32_t pix;               // first of the pixel's three accumulators: 3 * (pixel of the unit), + 3*P if the item belongs to slot 1
    uint32_t s, s_end;          // current sample and end of the run
    float band;                 // lib.rs:225-226 reflectance of the column
};

// One of the two units a wavefront has open (all members wave-uniform).
struct Slot {
    bool open;                  // a unit is assigned and not yet written out
    uint32_t left;              // its items not yet completed
    uint32_t x0, ly0, y0;       // first pixel: column, local row (output buffer), global row (camera)
};

RT_DEV Slot open_unit(const RtLaunch& L, uint32_t unit, uint32_t n_items) {
    Slot u;
    if (L.units_bottom_up) unit = L.n_units - 1u - unit;         // the launch's last rows are handed out first
    const uint32_t ux = unit % L.units_x, uy = unit / L.units_x;
    u.x0 = ux * L.unit_w;
    u.ly0 = uy * L.unit_h;
    const uint32_t k = u.ly0 / L.tile_rows;                      // unit_h divides tile_rows: one row tile per unit
    u.y0 = L.row_base + k * L.tile_stride_rows + (u.ly0 - k * L.tile_rows);
    u.left = n_items;
    u.open = true;
    return u;
}

// lanes 0..P-1 finish one pixel of the unit each (mean, gamma, RGBA8) and clear its accumulators
// `fold` (fold-order mode of the queue kernels): the unit's finished samples, [P][samples_per_pixel] colours in
// this wavefront's workspace; they are added here in sample order in fp32, the reference's fold (ray.cu:123-129,
// lib.rs:228-231), whatever order the lanes finished them in.
RT_DEV void write_unit(const RtLaunch& L, const Slot& u, unsigned long long* acc, uint32_t lane, bool exact_sum,
                       uint32_t* __restrict__ rgba, float4* __restrict__ linear, const float4* fold = nullptr) {
    const uint32_t P = L.unit_w * L.unit_h;
    if (fold) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the lanes' stores of the samples are visible
    if (lane < P) {
        const uint32_t dx = lane & (L.unit_w - 1u), dy = lane >> L.unit_shift;
        const uint32_t x = u.x0 + dx, ly = u.ly0 + dy, y = u.y0 + dy;
        V3 mean;
        if (fold) {
            V3 total = v3(0.0f, 0.0f, 0.0f);
            const uint32_t n = L.max_depth > 0u ? L.samples_per_pixel : 0u;
            const float4* mine = fold + (size_t)lane * L.samples_per_pixel;
            if (x < L.width && ly < L.rows_local && y < L.height)
                for (uint32_t s = 0u; s < n; s++) {
                    const float4 c = mine[s];
                    total = vadd(total, v3(c.x, c.y, c.z));      // ray.cu:126, lib.rs:230
                }
            const float fs = (float)L.samples_per_pixel;         // ray.cu:129, lib.rs:231
            mean = v3(total.x / fs, total.y / fs, total.z / fs);
        } else if (exact_sum) {
            mean = v3(fixed_mean((long long)acc[3u * lane], L.samples_per_pixel),
                      fixed_mean((long long)acc[3u * lane + 1u], L.samples_per_pixel),
                      fixed_mean((long long)acc[3u * lane + 2u], L.samples_per_pixel));
        } else {
            const float fs = (float)L.samples_per_pixel;         // ray.cu:129, lib.rs:231
            mean = v3(__uint_as_float((uint32_t)acc[3u * lane]) / fs, __uint_as_float((uint32_t)acc[3u * lane + 1u]) / fs,
                      __uint_as_float((uint32_t)acc[3u * lane + 2u]) / fs);
        }
        acc[3u * lane] = 0ull; acc[3u * lane + 1u] = 0ull; acc[3u * lane + 2u] = 0ull;
        if (x < L.width && ly < L.rows_local && y < L.height) {
            const size_t k = (size_t)ly * L.width + x;
            rgba[k] = quant(mean.x, L.flags) | (quant(mean.y, L.flags) << 8) | (quant(mean.z, L.flags) << 16) | (255u << 24);
            if (linear) linear[k] = make_float4(mean.x, mean.y, mean.z, 1.0f);
        }
    }
}

template <bool MULTI>
RT_DEV void render_body(const Scene& sc, uint32_t* __restrict__ rgba, float4* __restrict__ linear,
                        unsigned long long* __restrict__ segments, uint32_t* __restrict__ work,
                        const RtCamera& cam, const RtLaunch& L) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: kept in a scalar register
    const uint64_t seed = ((uint64_t)L.seed_hi << 32) | L.seed_lo;
    const uint32_t flags = L.flags;
    const uint32_t P = L.unit_w * L.unit_h;
    const uint32_t n_items = L.max_depth > 0u ? P * L.n_chunks : 0u;   // per unit
    const bool exact_sum = L.n_chunks > 1u;

    Tile tile = {rt_lds, rt_lds + tile_capacity(L), nullptr, orig_of(L, sc.mat())};
    const V3 T = ld3(L.translate);
    // per wavefront and slot: [P][3] 64-bit words, fixed-point sums (exact_sum) or the fp32 run sum bits
    unsigned long long* accum = accum_base(L) + wave * (6u * P);
    for (uint32_t k = lane; k < 6u * P; k += 64u) accum[k] = 0ull;
    if (!MULTI) {
        tile = stage_tile(L, sc.geom, sc.exact(), orig_of(L, sc.mat()), 0u, L.n_padded);
        __syncthreads();
    }

    Cand cand = cand_list(L);
    Stats st;
    Item it;
    it.x = it.y = it.pixel = it.pix = 0u; it.s = it.s_end = 0u; it.band = 0.0f;
    Slot u0, u1;
    u0.open = u1.open = false;
    u0.left = u1.left = 0u;
    u0.x0 = u0.ly0 = u0.y0 = u1.x0 = u1.ly0 = u1.y0 = 0u;
    uint32_t cur = 1u;                                           // slot items are handed out from
    uint32_t next = n_items;                                     // its next unclaimed item; n_items: none left (or none open yet)
    bool drained = false;                                        // the launch's unit counter has run out
    bool need = true, active = false, new_path = false;
    uint32_t seg = 0u, nb = 0u, nseg = 0u;
    V3 run = v3(0.0f, 0.0f, 0.0f), thr = v3(1.0f, 1.0f, 1.0f);
    V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, 1.0f);

    for (;;) {
        // completed units: write their pixels, free the slot
        if (u0.open && u0.left == 0u) { write_unit(L, u0, accum, lane, exact_sum, rgba, linear); u0.open = false; }
        if (u1.open && u1.left == 0u) { write_unit(L, u1, accum + 3u * P, lane, exact_sum, rgba, linear); u1.open = false; }

        const unsigned long long want = __ballot(need);
        if (want != 0ull) {
            // current unit handed out completely: open the next one in the other slot, once that is free
            if (next >= n_items && !drained && !(cur ? u0.open : u1.open)) {
                uint32_t unit = 0u;
                if (lane == 0u) unit = atomicAdd(work, 1u);
                unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
                if (unit >= L.n_units) {
                    drained = true;
                } else {
                    cur ^= 1u;
                    next = 0u;
                    if (cur) u1 = open_unit(L, unit, n_items); else u0 = open_unit(L, unit, n_items);
                }
            }
            if (next < n_items) {
                // the lanes that need an item take consecutive indices
                bool skipped = false;
                if (need) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                    const uint32_t item = next + rank;
                    if (item < n_items) {
                        const uint32_t pix = item & (P - 1u);
                        const uint32_t run_idx = item >> L.unit_pshift;
                        const uint32_t dx = pix & (L.unit_w - 1u), dy = pix >> L.unit_shift;
                        const uint32_t ly = (cur ? u1.ly0 : u0.ly0) + dy;
                        it.x = (cur ? u1.x0 : u0.x0) + dx;
                        it.y = (cur ? u1.y0 : u0.y0) + dy;
                        it.pix = 3u * pix + (cur ? 3u * P : 0u);
                        it.pixel = it.y * L.width + it.x;
                        it.s = run_idx * L.chunk;
                        it.s_end = it.s + L.chunk < L.samples_per_pixel ? it.s + L.chunk : L.samples_per_pixel;
                        if (flags & RT_FLAG_BAND_ALBEDO) {       // lib.rs:225-226
                            uint32_t gi = L.width / 5u;
                            it.band = 0.1f + 0.2f * (float)(it.x / (gi ? gi : 1u));
                        }
                        if (it.x < L.width && ly < L.rows_local && it.y < L.height && it.s < it.s_end) {
                            need = false;
                            active = true;
                            new_path = true;
                            run = v3(0.0f, 0.0f, 0.0f);
                        } else {
                            skipped = true;                      // pixel outside the image: nothing to trace
                        }
                    }
                }
                const uint32_t taken = (uint32_t)__popcll(want);
                const uint32_t given = taken < n_items - next ? taken : n_items - next;
                next += given;
                const uint32_t n_skipped = (uint32_t)__popcll(__ballot(skipped));
                if (cur) u1.left -= n_skipped; else u0.left -= n_skipped;
            }
        }
        // one place generates camera rays: for the first sample of a new item and for the next sample of a running one
        if (new_path) {
            float fx_unused, fy_unused;
            sample_ray(cam, L, seed, it.x, it.y, it.pixel, it.s, o, d, fx_unused, fy_unused);
            new_path = false;
        }
        st.lap(14);
        const bool more = active || !drained || next < n_items || u0.open || u1.open;
        if (MULTI) {
            if (!__syncthreads_or(more ? 1 : 0)) break;
        } else {
            if (__ballot(more) == 0ull) break;
        }
        Best b;
        b.t = L.t_max;
        b.idx = -1;
        const float a = vdot(d, d);                              // hoisted d.d (library.cu:201)
        if (MULTI) {
            for (uint32_t first = 0u; first < L.n_padded; first += RT_TILE_SPHERES) {
                uint32_t count = L.n_padded - first;
                if (count > RT_TILE_SPHERES) count = RT_TILE_SPHERES;
                if (first) __syncthreads();                      // previous tile fully scanned
                tile = stage_tile(L, sc.geom, sc.exact(), orig_of(L, sc.mat()), first, count);
                __syncthreads();
                if (active) scan_tile<false>(tile, count, (int)first, T, L.scan_scale, o, d, a, L.t_min, b, cand, st);
            }
        } else {
            if (active) scan_tile<false>(tile, L.n_padded, 0, T, L.scan_scale, o, d, a, L.t_min, b, cand, st);
        }
        bool finished = false;                                   // this lane completed its item in this round
        if (active) {
            nseg++;
            bool path_done;
            V3 color;
            if (b.idx >= 0) {                                    // ray.cu:35-40
                const float4 cr = sc.shade()[2 * b.idx];
                const float4 al = sc.shade()[2 * b.idx + 1];
                const uint32_t mat = sc.mat()[b.idx];
                HitInfo h = finish_hit(cr, b.t, o, d, flags);
                V3 nd = scatter(mat, al.w, sc.shade() + 2u * L.n_padded + (uint32_t)b.idx, d, h, seed, it.pixel, it.s, seg + 1u, flags, st);
                o = h.p;
                d = nd;
                if (flags & RT_FLAG_BAND_ALBEDO) {
                    nb++;
                } else {
                    thr = v3(thr.x * al.x, thr.y * al.y, thr.z * al.z);   // ray.cu:37
                }
                seg++;
                path_done = (seg == L.max_depth);                // ray.cu:30-32, lib.rs:190-192
                color = v3(0.0f, 0.0f, 0.0f);
                st.lap(12);
            } else {
                if (flags & RT_FLAG_BAND_ALBEDO) {               // lib.rs:195-205, innermost call outwards
                    color = sky(d, it.band);
                    for (uint32_t k = 0u; k < nb; k++) color = v3(it.band * color.x, it.band * color.y, it.band * color.z);
                } else {
                    V3 sk = sky(d, 0.5f);
                    color = v3(thr.x * sk.x, thr.y * sk.y, thr.z * sk.z);  // ray.cu:45
                }
                path_done = true;
            }
            if (path_done) {
                run = vadd(run, color);                          // ray.cu:126, lib.rs:230
                it.s++;
                seg = 0u;
                nb = 0u;
                thr = v3(1.0f, 1.0f, 1.0f);
                if (it.s < it.s_end) {
                    new_path = true;                             // camera ray at the top of the next trip
                } else {                                         // run complete: hand its sum to the pixel
                    if (exact_sum) {
                        atomicAdd(accum + it.pix, (unsigned long long)to_fixed(run.x));
                        atomicAdd(accum + it.pix + 1u, (unsigned long long)to_fixed(run.y));
                        atomicAdd(accum + it.pix + 2u, (unsigned long long)to_fixed(run.z));
                    } else {
                        accum[it.pix] = __float_as_uint(run.x);
                        accum[it.pix + 1u] = __float_as_uint(run.y);
                        accum[it.pix + 2u] = __float_as_uint(run.z);
                    }
                    active = false;
                    need = true;
                    finished = true;
                }
            }
        }
        st.lap(13);
        const unsigned long long fin = __ballot(finished);
        if (fin != 0ull) {
            const uint32_t n1 = (uint32_t)__popcll(__ballot(finished && it.pix >= 3u * P));
            u1.left -= n1;
            u0.left -= (uint32_t)__popcll(fin) - n1;
        }
    }

    if (segments) {
        // one 64-bit atomic per wavefront
        unsigned long long n64 = nseg;
        for (int off = 32; off > 0; off >>= 1) n64 += __shfl_xor(n64, off);
        if (lane == 0u && n64) atomicAdd(segments, n64);
#ifdef RT_STATS
        for (int w = 1; w < 8; w++) {
            unsigned long long x = st.v[w];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (lane == 0u && x) atomicAdd(segments + w, x);
        }
        for (int w = 0; w < 7; w++)
            if (lane == 0u) atomicAdd(segments + 9 + w, st.cyc[w]);
#endif
    }
}

// ---------------------------------------------------------------- render, camera rays traced in batches
// Used when every sample is its own work item (sample_chunk = 1) and the sphere list fits the LDS tile.
// A third of all path segments are camera rays, and those share their origin: the wavefront traces them 64
// at a time with scan_primary (4 operations per sphere instead of 9, every lane busy), shades them on the
// spot -- a miss is a finished sample and goes straight to the pixel's accumulator -- and queues the
// scattered rays of the hits in LDS.  The loop of render_body then runs on bounce segments only: a lane
// whose path has ended takes the next queued ray; when the queue is empty the next batch of camera rays
// of the unit is traced.  Image and segment count are those of render_body with sample_chunk = 1, bit for bit.
//   LDS: [geo S float4][geo0 S float4: this launch's camera-ray terms, see rt_camera_terms][block boxes 2 x L.scan_blocks float4] then per wavefront
//        [candidate lists RT_Q_CAND_CAP x 64 u16][accumulators 2 x P x 3 u64][queue RT_Q_WORDS x 64 u32]
// RT_Q_WORDS (rt_device.h) = 8 words per queued ray: origin xyz, direction xyz, sample,
// (slot * 64 + pixel of unit) << 16 | sphere hit

struct Path { V3 o, d, thr; uint32_t seg, nb; };

// Shades the segment that ended in `b`: scatter at the hit (ray.cu:35-40) or sky colour (ray.cu:43-45,
// lib.rs:195-205).  Returns true when the path is over, with its colour.
template <class SC>
RT_DEV bool shade_segment(const SC& sc, const RtLaunch& L, uint64_t seed, const Best& b, uint32_t pixel, uint32_t s,
                          float band, Path& p, V3& color, Stats& st) {
    const uint32_t flags = L.flags;
    if (b.idx >= 0) {
        const float4 cr = sc.shade()[2 * b.idx];
        const float4 al = sc.shade()[2 * b.idx + 1];
        const uint32_t mat = sc.mat()[b.idx];
        HitInfo h = finish_hit(cr, b.t, p.o, p.d, flags);
        V3 nd = scatter(mat, al.w, sc.shade() + 2u * L.n_padded + (uint32_t)b.idx, p.d, h, seed, pixel, s, p.seg + 1u, flags, st);
        p.o = h.p;
        p.d = nd;
        if (flags & RT_FLAG_BAND_ALBEDO) p.nb++;
        else p.thr = v3(p.thr.x * al.x, p.thr.y * al.y, p.thr.z * al.z);
        p.seg++;
        color = v3(0.0f, 0.0f, 0.0f);
        st.lap(12);
        return p.seg == L.max_depth;
    }
    if (flags & RT_FLAG_BAND_ALBEDO) {
        color = sky(p.d, band);
        for (uint32_t k = 0u; k < p.nb; k++) color = v3(band * color.x, band * color.y, band * color.z);
    } else {
        V3 sk = sky(p.d, 0.5f);
        color = v3(p.thr.x * sk.x, p.thr.y * sk.y, p.thr.z * sk.z);
    }
    st.lap(13);
    return true;
}

RT_DEV float band_of(const RtLaunch& L, uint32_t x) {              // lib.rs:225-226
    uint32_t gi = L.width / 5u;
    return 0.1f + 0.2f * (float)(x / (gi ? gi : 1u));
}

// One finished sample.  acc = the wavefront's accumulators ([slot][pixel][r, g, b]), pixslot = 3 * pixel of the
// unit (+ 3P for slot 1): the index of the pixel's first accumulator.
// Exact mode: added to the pixel's fixed-point accumulators.  Fold-order mode (fold != NULL): stored at
// [slot][pixel][sample] of the wavefront's workspace, to be added in sample order when the unit is complete.
RT_DEV void add_sample(unsigned long long* acc, uint32_t P, uint32_t pixslot, uint32_t s, uint32_t spp, V3 color, float4* fold) {
    if (fold) {
        const uint32_t sp = (pixslot * 43691u) >> 17;             // pixslot / 3 = slot * P + pixel (exact below 98,304)
        fold[(size_t)sp * spp + s] = make_float4(color.x, color.y, color.z, 0.0f);
        return;
    }
    atomicAdd(acc + pixslot, (unsigned long long)to_fixed(color.x));
    atomicAdd(acc + pixslot + 1u, (unsigned long long)to_fixed(color.y));
    atomicAdd(acc + pixslot + 2u, (unsigned long long)to_fixed(color.z));
}

// Queue entry j becomes the lane's path: the ray that left the camera ray's hit, one segment done.
template <class SC>
RT_DEV void take_queued(const SC& sc, const RtLaunch& L, const uint32_t* q, uint32_t j, const Slot& u0, const Slot& u1,
                        Path& p, uint32_t& s, uint32_t& pixel, uint32_t& pixslot, float& band) {
    const uint32_t P = L.unit_w * L.unit_h;
    p.o = v3(__uint_as_float(q[j]), __uint_as_float(q[64u + j]), __uint_as_float(q[128u + j]));
    p.d = v3(__uint_as_float(q[192u + j]), __uint_as_float(q[256u + j]), __uint_as_float(q[320u + j]));
    s = q[384u + j];
    const uint32_t meta = q[448u + j];
    const uint32_t sphere = meta & 0xffffu, ps = meta >> 16;
    const bool in1 = ps >= 64u;
    const uint32_t pix = ps & 63u;
    const uint32_t dx = pix & (L.unit_w - 1u), dy = pix >> L.unit_shift;
    const uint32_t x = (in1 ? u1.x0 : u0.x0) + dx, y = (in1 ? u1.y0 : u0.y0) + dy;
    pixel = y * L.width + x;
    pixslot = 3u * pix + (in1 ? 3u * P : 0u);
    p.seg = 1u;
    if (L.flags & RT_FLAG_BAND_ALBEDO) {
        band = band_of(L, x);
        p.nb = 1u;
    } else {
        const float4 al = sc.shade()[2u * sphere + 1u];
        p.thr = v3(al.x, al.y, al.z);                            // 1 * albedo (ray.cu:37)
        p.nb = 0u;
    }
}

// Appends the lanes' scattered camera rays (`cont`) to the empty queue, packed from slot 0; returns their number.
RT_DEV uint32_t queue_rays(uint32_t* q, bool cont, const Path& p, uint32_t s, uint32_t slot, uint32_t pix, uint32_t sphere) {
    const unsigned long long cm = __ballot(cont);
    if (cont) {
        const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
        q[r] = __float_as_uint(p.o.x); q[64u + r] = __float_as_uint(p.o.y); q[128u + r] = __float_as_uint(p.o.z);
        q[192u + r] = __float_as_uint(p.d.x); q[256u + r] = __float_as_uint(p.d.y); q[320u + r] = __float_as_uint(p.d.z);
        q[384u + r] = s;
        q[448u + r] = ((slot * 64u + pix) << 16) | sphere;
    }
    return (uint32_t)__popcll(cm);
}

// Traces and shades the camera rays of samples first .. first+63 of unit `u` (sample i: pixel i % P, sample
// number i / P).  Called by the whole wavefront.  Finished samples (miss, depth limit, pixel outside the
// image) are counted in `completed`; the others leave a ray in the queue, packed from slot 0 (`queued`).
// LISTS = false: a build of the caller without the units' sphere lists (ul, ul_k unused).
template <bool LISTS, class SC>
RT_DEV void primary_batch(const SC& sc, const Tile& tile, const float4* __restrict__ geo0, const RtCamera& cam,
                          const RtLaunch& L, uint64_t seed, V3 T, const Slot& u, uint32_t slot, uint32_t first,
                          uint32_t n_samp, unsigned long long* acc, float4* fold, uint32_t* q, Cand& cand,
                          const UnitList& ul, uint32_t ul_k, uint32_t& queued, uint32_t& completed, uint32_t& nseg, Stats& st) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t P = L.unit_w * L.unit_h;
    const uint32_t sidx = first + lane;
    const uint32_t pix = sidx & (P - 1u), s = sidx >> L.unit_pshift;
    const uint32_t dx = pix & (L.unit_w - 1u), dy = pix >> L.unit_shift;
    const uint32_t x = u.x0 + dx, ly = u.ly0 + dy, y = u.y0 + dy;
    const bool in_unit = sidx < n_samp;
    const bool traced = in_unit && x < L.width && ly < L.rows_local && y < L.height;
    bool done = in_unit && !traced, cont = false;
    Path p;
    p.o = p.d = v3(0.0f, 0.0f, 0.0f);
    uint32_t sphere = 0u;
    if (traced) {
        const uint32_t pixel = y * L.width + x;
        const float band = (L.flags & RT_FLAG_BAND_ALBEDO) ? band_of(L, x) : 0.0f;
        float fx, fy;
        sample_ray(cam, L, seed, x, y, pixel, s, p.o, p.d, fx, fy);
        p.thr = v3(1.0f, 1.0f, 1.0f);
        p.seg = 0u;
        p.nb = 0u;
        Best b;
        b.t = L.t_max;
        b.idx = -1;
        st.lap(14);
        // ul_k (wave-uniform): entries of the unit's list, or RT_UL_NONE: the whole scene
        const bool listed = LISTS && ul_k != RT_UL_NONE;
        const float4* terms = listed ? ul.terms : geo0;
        const uint16_t* map = listed ? ul.idx : nullptr;
        const uint32_t n_terms = listed ? ul_k : L.n_padded;
        scan_primary(terms, map, tile, n_terms, 0, fx, fy, p.o, p.d, vdot(p.d, p.d), L.t_min, b, cand, st);
#ifdef RT_EXP_PRIMARY2
        scan_primary(terms, map, tile, n_terms, 0, fx, fy, p.o, p.d, vdot(p.d, p.d), L.t_min, b, cand, st);   // cost experiment: changes nothing (strict bound)
#endif
        nseg++;
        sphere = (uint32_t)b.idx;
        V3 color;
        if (shade_segment(sc, L, seed, b, pixel, s, band, p, color, st)) {
            add_sample(acc, P, 3u * pix + slot * 3u * P, s, L.samples_per_pixel, color, fold);
            done = true;
        } else {
            cont = true;
        }
    }
    queued = queue_rays(q, cont, p, s, slot, pix, sphere);
    completed = (uint32_t)__popcll(__ballot(done));
}

// BOXES: built with the blocks' bounding boxes in the bounce scan (they cost registers also where a launch has none).
template <bool FOLD, bool BOXES>
RT_DEV void render_queue_body(const Scene& sc, uint32_t* __restrict__ rgba, float4* __restrict__ linear,
                              unsigned long long* __restrict__ segments, uint32_t* __restrict__ work,
                              float4* __restrict__ fold_all, const float4* __restrict__ camterms, const RtCamera& cam,
                              const RtLaunch& L) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: kept in a scalar register
    const uint64_t seed = ((uint64_t)L.seed_hi << 32) | L.seed_lo;
    const uint32_t flags = L.flags;
    const uint32_t P = L.unit_w * L.unit_h;
    const uint32_t n_samp = L.max_depth > 0u ? P * L.samples_per_pixel : 0u;   // per unit
    const uint32_t S = L.n_padded;
    const V3 T = ld3(L.translate);

    float4* geo = rt_lds;
    float4* geo0 = rt_lds + S;
    float4* box = geo0 + S;                                       // the blocks' boxes (scan_tile), 2 x L.scan_blocks
    char* mine = reinterpret_cast<char*>(box + 2u * L.scan_blocks) + wave * (RT_Q_CAND_CAP * 128u + 48u * P + RT_Q_WORDS * 256u + RT_UL_BYTES(L.ulist_cap));
    Cand cand;
    cand.col = reinterpret_cast<uint16_t*>(mine) + lane;
    cand.n = 0u;
    cand.cap = RT_Q_CAND_CAP;
    unsigned long long* accum = reinterpret_cast<unsigned long long*>(mine + RT_Q_CAND_CAP * 128u);
    uint32_t* q = reinterpret_cast<uint32_t*>(mine + RT_Q_CAND_CAP * 128u + 48u * P);
    // the list of the unit whose camera rays are being traced (unit_list)
    UnitList ul;
    ul.terms = reinterpret_cast<float4*>(mine + RT_Q_CAND_CAP * 128u + 48u * P + RT_Q_WORDS * 256u);
    ul.idx = reinterpret_cast<uint16_t*>(ul.terms + L.ulist_cap);
    uint32_t ul_k = RT_UL_NONE;
    for (uint32_t k = lane; k < 6u * P; k += 64u) accum[k] = 0ull;
    // fold-order mode: this wavefront's [2 slots][P][samples_per_pixel] colours in the launch's workspace
    float4* fold = FOLD ? fold_all + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * (2u * P * (size_t)L.samples_per_pixel) : nullptr;
    const float4* fold1 = FOLD ? fold + (size_t)P * L.samples_per_pixel : nullptr;
    // stage the spheres: the filter data of scan_tile and the camera-ray terms of this launch (rt_camera_terms)
    for (uint32_t i = threadIdx.x; i < S; i += blockDim.x) {
        geo[i] = sc.geom[i];
        geo0[i] = camterms[i];
    }
    for (uint32_t i = threadIdx.x; i < 2u * L.scan_blocks; i += blockDim.x) box[i] = sc.geom[S + i];
    __syncthreads();
    // the exact-test data stays in global memory (8 KB, read a couple of times per ray, cached): with it in LDS
    // a compute unit holds four workgroups instead of five, measured 2 % slower on the 485-sphere scene
    const Tile tile = {geo, sc.exact(), L.scan_blocks ? box : nullptr, orig_of(L, sc.mat())};

    Stats st;
    Slot u0, u1;
    u0.open = u1.open = false;
    u0.left = u1.left = 0u;
    u0.x0 = u0.ly0 = u0.y0 = u1.x0 = u1.ly0 = u1.y0 = 0u;
    uint32_t cur = 1u;                 // slot whose camera rays are being traced
    uint32_t next_samp = n_samp;       // its next untraced sample; n_samp: none left (or nothing open yet)
    bool drained = false;              // the launch's unit counter has run out
    uint32_t q_head = 0u, q_count = 0u;
    bool need = true, active = false;
    Path p;
    p.o = v3(0.0f, 0.0f, 0.0f); p.d = v3(0.0f, 0.0f, 1.0f); p.thr = v3(1.0f, 1.0f, 1.0f); p.seg = 0u; p.nb = 0u;
    uint32_t pixel = 0u, s = 0u, pixslot = 0u, nseg = 0u;
    float band = 0.0f;

    for (;;) {
        // completed units: write their pixels, free the slot
        if (u0.open && u0.left == 0u) { write_unit(L, u0, accum, lane, true, rgba, linear, fold); u0.open = false; }
        if (u1.open && u1.left == 0u) { write_unit(L, u1, accum + 3u * P, lane, true, rgba, linear, fold1); u1.open = false; }

        // lanes without a path take queued rays
        unsigned long long want = __ballot(need);
        if (want != 0ull && q_head < q_count) {
            const uint32_t j = q_head + __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
            if (need && j < q_count) {
                take_queued(sc, L, q, j, u0, u1, p, s, pixel, pixslot, band);
                need = false;
                active = true;
            }
            const uint32_t asked = (uint32_t)__popcll(want);
            q_head += asked < q_count - q_head ? asked : q_count - q_head;
            want = __ballot(need);
        }
        // lanes still without one (the queue is empty now): trace the next camera rays, then come round again
        if (want != 0ull) {
            bool can = true;
            if (next_samp >= n_samp) {               // this unit's camera rays are all traced: open the next unit
                if (drained || (cur ? u0.open : u1.open)) {
                    can = false;
                } else {
                    uint32_t unit = 0u;
                    if (lane == 0u) unit = atomicAdd(work, 1u);
                    unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
                    if (unit >= L.n_units) {
                        drained = true;
                        can = false;
                    } else {
                        cur ^= 1u;
                        next_samp = 0u;
                        if (cur) u1 = open_unit(L, unit, n_samp); else u0 = open_unit(L, unit, n_samp);
                        if (n_samp == 0u) continue;
                        ul_k = L.ulist_cap ? uniform_u32(unit_list(sc.geom, sc.exact(), geo0, cam, L, cur ? u1.x0 : u0.x0, cur ? u1.y0 : u0.y0, ul, lane)) : RT_UL_NONE;
                    }
                }
            }
            if (can) {
                uint32_t completed = 0u;
                primary_batch<true>(sc, tile, geo0, cam, L, seed, T, cur ? u1 : u0, cur, next_samp, n_samp,
                                    accum, fold, q, cand, ul, ul_k, q_count, completed, nseg, st);
                q_head = 0u;
                next_samp = next_samp + 64u < n_samp ? next_samp + 64u : n_samp;
                if (cur) u1.left -= completed; else u0.left -= completed;
                continue;
            }
        }

        const bool more = active || q_head < q_count || !drained || next_samp < n_samp || u0.open || u1.open;
        if (__ballot(more) == 0ull) break;

        bool finished = false;                                   // this lane's path ended in this round
        st.lap(14);
        Best b;
        b.t = L.t_max;
        b.idx = -1;
        if (BOXES) scan_tile<true, true>(tile, S, 0, T, L.scan_scale, p.o, p.d, vdot(p.d, p.d), L.t_min, b, cand, st, active);   // every lane: see scan_tile
        else if (active) scan_tile<true, false>(tile, S, 0, T, L.scan_scale, p.o, p.d, vdot(p.d, p.d), L.t_min, b, cand, st);
        if (active) {
            nseg++;
            V3 color;
            if (shade_segment(sc, L, seed, b, pixel, s, band, p, color, st)) {
                add_sample(accum, P, pixslot, s, L.samples_per_pixel, color, fold);
                active = false;
                need = true;
                finished = true;
            }
        }
        const unsigned long long fin = __ballot(finished);
        if (fin != 0ull) {
            const uint32_t n1 = (uint32_t)__popcll(__ballot(finished && pixslot >= 3u * P));
            u1.left -= n1;
            u0.left -= (uint32_t)__popcll(fin) - n1;
        }
    }

    if (segments) {
        unsigned long long n64 = nseg;
        for (int off = 32; off > 0; off >>= 1) n64 += __shfl_xor(n64, off);
        if (lane == 0u && n64) atomicAdd(segments, n64);
#ifdef RT_STATS
        for (int w = 1; w < 8; w++) {
            unsigned long long x = st.v[w];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (lane == 0u && x) atomicAdd(segments + w, x);
        }
        for (int w = 0; w < 7; w++)
            if (lane == 0u) atomicAdd(segments + 9 + w, st.cyc[w]);
#endif
    }
}

// ---------------------------------------------------------------- render, units shared by the workgroup
// The same path tracing as render_queue_body (exact accumulation), with the pixel units held by the WORKGROUP instead
// of by each wavefront: a pool of L.pool_slots open units in LDS, from whose current unit every wavefront of the
// workgroup takes its camera-ray batches (one LDS atomic per batch).  What this buys: (1) a unit's last, longest paths
// hold up one pool slot, not the wavefront that traced them -- a wavefront never waits for its own stragglers before it
// may start new camera rays; (2) at the end of the launch the workgroup's wavefronts finish the last units together
// instead of each finishing two of its own.  The image is the same bit for bit: sums are exact and order-free.
//   slot (8 words): x0, y0 (global row), ly0 (local row), next sample to trace, samples not yet finished, open flag,
//        entries of the unit's list (unit_list; RT_UL_NONE: none)
//   ctl: [0] slot the camera rays are taken from (POOL_NONE before the first unit, POOL_DRAINED when the launch's
//        counter has run out), [1] lock of the wavefront that is opening a unit
//   LDS: [geo S float4][geo0 S float4][block boxes 2 x L.scan_blocks float4][slots NS x 8 u32][ctl 4 u32][accumulators NS x P x 3 u64][unit lists NS x
//        RT_UL_BYTES] then per wavefront [candidate lists RT_Q_CAND_CAP x 64 u16][queue RT_Q_WORDS x 64 u32]
// A finished sample is added to its pixel's accumulators and then taken off its slot's count (LDS operations of a
// wavefront are performed in order); whoever takes the last one off writes the unit's pixels and frees the slot.
#define POOL_NONE 0xffffffffu
#define POOL_DRAINED 0xfffffffeu

// Opens the workgroup's next unit and makes it the one camera rays are taken from, if nobody else is at it.
// ahead = POOL_NONE: the caller found the current unit used up (or none there yet).  ahead = a slot: the caller holds
// a batch of that slot's unit, which is about to be used up: the next unit is opened early, so that the wavefronts
// do not find themselves without camera rays for the time the launch's counter takes to answer; wavefronts go on
// taking the older unit's last batches from the slot they remember (pool_next_batch).
// Returns true if it makes sense to look at the control word again right away.
// The list of pool slot `slot` (the workgroup's LDS layout: see render_pool_body).
RT_DEV UnitList pool_list(const RtLaunch& L, uint32_t slot) {
    char* base = reinterpret_cast<char*>(rt_lds + 2u * L.n_padded + 2u * L.scan_blocks) + L.pool_slots * (32u + 24u * L.unit_w * L.unit_h) + 16u;
    UnitList ul;
    ul.terms = reinterpret_cast<float4*>(base + slot * RT_UL_BYTES(L.ulist_cap));
    ul.idx = reinterpret_cast<uint16_t*>(ul.terms + L.ulist_cap);
    return ul;
}

template <bool LISTS>
RT_DEV bool pool_open(const RtLaunch& L, uint32_t* slots, uint32_t* ctl, uint32_t* __restrict__ work, uint32_t n_samp,
                      uint32_t lane, uint32_t ahead, const RtCamera& cam, const float4* __restrict__ geom, const float4* __restrict__ exact, bool& drained) {
    uint32_t lock = 1u;
    if (lane == 0u) lock = atomicCAS(&ctl[1], 0u, 1u);
    if (uniform_u32(lock) != 0u) return false;                    // somebody is opening one
    // With the lock held nobody else opens a unit.
    const uint32_t c = uniform_u32(__hip_atomic_load(&ctl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
    bool go;
    if (ahead != POOL_NONE) {
        go = c == ahead;          // still the newest (it cannot have changed hands: the caller holds a batch of it)
    } else {
        // If one was opened in the meantime (possibly in the very slot the caller saw, freed and taken again), its
        // samples come first.
        go = c == POOL_NONE || uniform_u32(__hip_atomic_load(&slots[c * 8u + 3u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= n_samp;
    }
    bool again = !go;
    if (go) {
        const bool is_free = lane < L.pool_slots && __hip_atomic_load(&slots[lane * 8u + 5u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u;
        const unsigned long long free_mask = __ballot(is_free);
        if (free_mask != 0ull) {                                  // else: every slot still has paths in flight
            const uint32_t f = (uint32_t)__ffsll((long long)free_mask) - 1u;
            uint32_t unit = 0u;
            if (lane == 0u) unit = atomicAdd(work, 1u);
            unit = uniform_u32(unit);
            if (unit >= L.n_units) {
                // no more units.  Opened early, the older unit's last batches are still out: its slot stays the
                // current one, and whoever finds it used up comes here again and sets the end mark.
                if (ahead == POOL_NONE) {
                    if (lane == 0u) __hip_atomic_store(&ctl[0], POOL_DRAINED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    drained = true;
                }
            } else {
                const Slot u = open_unit(L, unit, n_samp);
                // the free slot's list is nobody's: written here, published with the slot below
                uint32_t ul_k = RT_UL_NONE;
                if (LISTS && L.ulist_cap) ul_k = uniform_u32(unit_list(geom, exact, rt_lds + L.n_padded, cam, L, u.x0, u.y0, pool_list(L, f), lane));
                if (lane == 0u) {
                    slots[f * 8u + 6u] = ul_k;
                    slots[f * 8u + 0u] = u.x0;
                    slots[f * 8u + 1u] = u.y0;
                    slots[f * 8u + 2u] = u.ly0;
                    slots[f * 8u + 4u] = n_samp;
                    slots[f * 8u + 5u] = 1u;
                    __hip_atomic_store(&slots[f * 8u + 3u], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&ctl[0], f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                again = true;
            }
        }
    }
    if (lane == 0u) __hip_atomic_store(&ctl[1], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    return again;
}

// Takes the workgroup's next camera-ray batch: slot, first sample and the slot's first words (x0, y0, ly0).  False:
// none to be had right now (the current unit is used up and another wavefront is opening the next one, every slot is
// still held by unfinished paths, or the launch's counter has run out: `drained`).
// ahead_at: the wavefront whose batch contains this sample of a unit opens the next unit (pool_open).
template <bool LISTS>
RT_DEV bool pool_next_batch(const RtLaunch& L, uint32_t* slots, uint32_t* ctl, uint32_t* __restrict__ work, uint32_t n_samp,
                            uint32_t ahead_at, uint32_t lane, const RtCamera& cam, const float4* __restrict__ geom, const float4* __restrict__ exact, uint32_t& hint, uint32_t& slot, uint32_t& first,
                            uint4& head, bool& drained) {
    bool got = false;
    // The slot this wavefront last took a batch from is tried first.  A unit that still has untraced samples is good
    // to take from whatever the control word says (a slot's counter is set to 0 only after its other fields, and
    // stays at or above the unit's samples once they are used up); the first failure drops the hint, which bounds
    // what waiting wavefronts add to a used-up counter.
    if (hint != POOL_NONE) {
        uint32_t old = 0u;
        if (lane == 0u) old = __hip_atomic_fetch_add(&slots[hint * 8u + 3u], 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // The slot's first words are read behind the atomic without waiting for it in between: the LDS performs a
        // wavefront's operations in order, and once the batch is this wavefront's the slot cannot change hands.
        asm volatile("" ::: "memory");
        head = *reinterpret_cast<const uint4*>(slots + hint * 8u);
        old = uniform_u32(old);
        if (old < n_samp) { slot = hint; first = old; got = true; }
        else hint = POOL_NONE;
    }
    for (int attempt = 0; attempt < 2 && !got; attempt++) {
        const uint32_t c = uniform_u32(__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (c == POOL_DRAINED) { drained = true; return false; }
        if (c != POOL_NONE) {
            // look before taking: a used-up unit's counter must not be pushed round its 32 bits by waiting wavefronts
            const uint32_t seen = uniform_u32(__hip_atomic_load(&slots[c * 8u + 3u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (seen < n_samp) {
                uint32_t old = 0u;
                if (lane == 0u) old = __hip_atomic_fetch_add(&slots[c * 8u + 3u], 64u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = uniform_u32(old);
                if (old < n_samp) {
                    slot = c; first = old; hint = c;
                    head = *reinterpret_cast<const uint4*>(slots + c * 8u);
                    got = true;
                    break;
                }
            }
        }
        // the current unit is used up (or there is none yet): open the next one, if nobody else is at it
        if (!pool_open<LISTS>(L, slots, ctl, work, n_samp, lane, POOL_NONE, cam, geom, exact, drained)) return false;
    }
    if (got && ahead_at >= first && ahead_at < first + 64u) {
        bool unused = false;
        (void)pool_open<LISTS>(L, slots, ctl, work, n_samp, lane, slot, cam, geom, exact, unused);
    }
    return got;
}

// Slot `sl`'s samples are all finished (wave-uniform call): writes the unit's pixels and frees the slot.
// fold (fold-order mode): the workgroup's [slot][pixel][sample] colours; they were stored by any of its wavefronts
// before those took the samples off the slot's count (release), and are read here after the count was seen (acquire).
RT_DEV void pool_write(const RtLaunch& L, uint32_t* slots, unsigned long long* accum, uint32_t sl, uint32_t lane,
                       uint32_t* __restrict__ rgba, float4* __restrict__ linear, const float4* fold) {
    const uint32_t P = L.unit_w * L.unit_h;
    Slot u;
    u.open = true; u.left = 0u;
    u.x0 = uniform_u32(slots[sl * 8u + 0u]);
    u.y0 = uniform_u32(slots[sl * 8u + 1u]);
    u.ly0 = uniform_u32(slots[sl * 8u + 2u]);
    write_unit(L, u, accum + sl * 3u * P, lane, true, rgba, linear, fold ? fold + (size_t)sl * P * L.samples_per_pixel : nullptr);
    if (lane == 0u) __hip_atomic_store(&slots[sl * 8u + 5u], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// `n` samples of slot `sl` are finished (wave-uniform call): takes them off the slot's count; if they were the last,
// writes the unit's pixels and frees the slot.
RT_DEV void pool_finish(const RtLaunch& L, uint32_t* slots, unsigned long long* accum, uint32_t sl, uint32_t n, uint32_t lane,
                        uint32_t* __restrict__ rgba, float4* __restrict__ linear, const float4* fold) {
    if (fold) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // every lane's stored samples before lane 0's count
    uint32_t old = 0u;
    if (lane == 0u) old = __hip_atomic_fetch_sub(&slots[sl * 8u + 4u], n, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (uniform_u32(old) == n) pool_write(L, slots, accum, sl, lane, rgba, linear, fold);
}

// LISTS, BOXES: built with the pixel units' sphere lists / the blocks' bounding boxes in the bounce scan (both cost
// registers also where a launch has none).
template <bool FOLD, bool LISTS, bool BOXES>
RT_DEV void render_pool_body(const Scene& sc, uint32_t* __restrict__ rgba, float4* __restrict__ linear,
                             unsigned long long* __restrict__ segments, uint32_t* __restrict__ work,
                             float4* __restrict__ fold_all, const float4* __restrict__ camterms, const RtCamera& cam,
                             const RtLaunch& L) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t seed = ((uint64_t)L.seed_hi << 32) | L.seed_lo;
    const uint32_t P = L.unit_w * L.unit_h;
    const uint32_t n_samp = P * L.samples_per_pixel;              // per unit (the launcher keeps max_depth = 0 off this kernel)
    const uint32_t S = L.n_padded, NS = L.pool_slots;
    const V3 T = ld3(L.translate);

    float4* geo = rt_lds;
    float4* geo0 = rt_lds + S;
    float4* box = geo0 + S;                                       // the blocks' boxes (scan_tile), 2 x L.scan_blocks
    uint32_t* slots = reinterpret_cast<uint32_t*>(box + 2u * L.scan_blocks);
    uint32_t* ctl = slots + NS * 8u;
    unsigned long long* accum = reinterpret_cast<unsigned long long*>(ctl + 4u);
    char* mine = reinterpret_cast<char*>(accum + NS * 3u * P) + NS * RT_UL_BYTES(L.ulist_cap) + wave * (RT_Q_CAND_CAP * 128u + RT_Q_WORDS * 256u);
    Cand cand;
    cand.col = reinterpret_cast<uint16_t*>(mine) + lane;
    cand.n = 0u;
    cand.cap = RT_Q_CAND_CAP;
    uint32_t* q = reinterpret_cast<uint32_t*>(mine + RT_Q_CAND_CAP * 128u);
    // fold-order mode: this workgroup's [NS slots][P][samples_per_pixel] colours in the launch's workspace
    float4* fold = FOLD ? fold_all + (size_t)blockIdx.x * ((size_t)NS * P * L.samples_per_pixel) : nullptr;
    for (uint32_t k = threadIdx.x; k < NS * 8u; k += blockDim.x) slots[k] = (k & 7u) == 3u ? 0x80000000u : 0u;   // closed: used up
    if (threadIdx.x == 0u) { ctl[0] = POOL_NONE; ctl[1] = 0u; }
    for (uint32_t k = threadIdx.x; k < NS * 3u * P; k += blockDim.x) accum[k] = 0ull;
    for (uint32_t i = threadIdx.x; i < S; i += blockDim.x) {
        geo[i] = sc.geom[i];
        geo0[i] = camterms[i];
    }
    for (uint32_t i = threadIdx.x; i < 2u * L.scan_blocks; i += blockDim.x) box[i] = sc.geom[S + i];
    __syncthreads();
    const Tile tile = {geo, sc.exact(), L.scan_blocks ? box : nullptr, orig_of(L, sc.mat())};

    Stats st;
    bool drained = false;
    uint32_t hint = POOL_NONE;
    // the next unit is opened when about one batch per wavefront is left of the current one
    const uint32_t ahead_at = n_samp > blockDim.x ? n_samp - blockDim.x : 0u;
    uint32_t q_head = 0u, q_count = 0u;
    bool need = true, active = false;
    Path p;
    p.o = v3(0.0f, 0.0f, 0.0f); p.d = v3(0.0f, 0.0f, 1.0f); p.thr = v3(1.0f, 1.0f, 1.0f); p.seg = 0u; p.nb = 0u;
    uint32_t pixel = 0u, s = 0u, pixslot = 0u, nseg = 0u;
    float band = 0.0f;

    for (;;) {
        // lanes without a path take queued rays
        unsigned long long want = __ballot(need);
        if (want != 0ull && q_head < q_count) {
            const uint32_t j = q_head + __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
            if (need && j < q_count) {
                p.o = v3(__uint_as_float(q[j]), __uint_as_float(q[64u + j]), __uint_as_float(q[128u + j]));
                p.d = v3(__uint_as_float(q[192u + j]), __uint_as_float(q[256u + j]), __uint_as_float(q[320u + j]));
                s = q[384u + j];
                const uint32_t meta = q[448u + j];
                const uint32_t sphere = meta & 0xffffu, ps = meta >> 16;
                const uint32_t sl = ps >> 6, pix = ps & 63u;
                const uint32_t dx = pix & (L.unit_w - 1u), dy = pix >> L.unit_shift;
                const uint2 at = *reinterpret_cast<const uint2*>(slots + sl * 8u);
                const uint32_t x = at.x + dx, y = at.y + dy;
                pixel = y * L.width + x;
                pixslot = 3u * pix + sl * 3u * P;
                p.seg = 1u;
                if (L.flags & RT_FLAG_BAND_ALBEDO) {
                    band = band_of(L, x);
                    p.nb = 1u;
                } else {
                    const float4 al = sc.shade()[2u * sphere + 1u];
                    p.thr = v3(al.x, al.y, al.z);                // 1 * albedo (ray.cu:37)
                    p.nb = 0u;
                }
                need = false;
                active = true;
            }
            const uint32_t asked = (uint32_t)__popcll(want);
            q_head += asked < q_count - q_head ? asked : q_count - q_head;
            want = __ballot(need);
        }
        // lanes still without one (the queue is empty now): trace the workgroup's next camera rays, then come round again
        if (want != 0ull && !drained) {
            uint32_t sl = 0u, first = 0u;
            uint4 head = make_uint4(0u, 0u, 0u, 0u);
            if (pool_next_batch<LISTS>(L, slots, ctl, work, n_samp, ahead_at, lane, cam, sc.geom, sc.exact(), hint, sl, first, head, drained)) {
                Slot u;
                u.open = true; u.left = 0u;
                u.x0 = uniform_u32(head.x);
                u.y0 = uniform_u32(head.y);
                u.ly0 = uniform_u32(head.z);
                uint32_t completed = 0u;
                // (the list's length sits in the slot like x0: once the batch is this wavefront's the slot cannot change hands)
                const uint32_t ul_k = LISTS ? uniform_u32(slots[sl * 8u + 6u]) : RT_UL_NONE;
                primary_batch<LISTS>(sc, tile, geo0, cam, L, seed, T, u, sl, first, n_samp, accum, fold, q, cand, pool_list(L, sl), ul_k,
                                     q_count, completed, nseg, st);
                q_head = 0u;
                if (completed) pool_finish(L, slots, accum, sl, completed, lane, rgba, linear, fold);
                continue;
            }
        }

        if (__ballot(active) == 0ull) {
            // nothing to trace: done if the launch has no more units (open ones are finished by whoever holds their paths);
            // otherwise wait a moment for a slot or for the wavefront that is opening the next unit
            if (drained) break;
            __builtin_amdgcn_s_sleep(4);
        }

        bool finished = false;                                   // this lane's path ended in this round
        st.lap(14);
        Best b;
        b.t = L.t_max;
        b.idx = -1;
        if (BOXES) scan_tile<true, true>(tile, S, 0, T, L.scan_scale, p.o, p.d, vdot(p.d, p.d), L.t_min, b, cand, st, active);   // every lane: see scan_tile
        else if (active) scan_tile<true, false>(tile, S, 0, T, L.scan_scale, p.o, p.d, vdot(p.d, p.d), L.t_min, b, cand, st);
        if (active) {
            nseg++;
            V3 color;
            if (shade_segment(sc, L, seed, b, pixel, s, band, p, color, st)) {
                add_sample(accum, P, pixslot, s, L.samples_per_pixel, color, fold);
                active = false;
                need = true;
                finished = true;
            }
        }
        // Finished samples come off their units' counts.  A wavefront's lanes hold paths of a few units: the first
        // finished lane of each unit takes all of that unit's off, all of them in one LDS instruction (an atomic per
        // lane on the same word would cost the LDS a cycle per lane); whoever takes a unit's last ones writes the unit.
        unsigned long long fin = __ballot(finished);
        if (fin != 0ull) {
            const uint32_t myslot = ((pixslot * 43691u) >> 17) >> L.unit_pshift;   // pixslot / 3 (exact below 98,304) / P
            uint32_t mine_n = 0u;                                 // > 0: this lane takes that many off its unit
            do {
                const uint32_t l0 = (uint32_t)__ffsll((long long)fin) - 1u;
                const uint32_t sl = (uint32_t)__builtin_amdgcn_readlane((int)myslot, (int)l0);
                const unsigned long long same = __ballot(finished && myslot == sl);
                if (lane == l0) mine_n = (uint32_t)__popcll(same);
                fin &= ~same;
            } while (fin != 0ull);
            if (FOLD) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the lanes' stored samples before the counts
            uint32_t before = 0u;
            if (mine_n != 0u) before = __hip_atomic_fetch_sub(&slots[myslot * 8u + 4u], mine_n, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            unsigned long long last = __ballot(mine_n != 0u && before == mine_n);
            while (last != 0ull) {
                const uint32_t l0 = (uint32_t)__ffsll((long long)last) - 1u;
                pool_write(L, slots, accum, (uint32_t)__builtin_amdgcn_readlane((int)myslot, (int)l0), lane, rgba, linear, fold);
                last &= last - 1ull;
            }
        }
    }

    if (segments) {
        unsigned long long n64 = nseg;
        for (int off = 32; off > 0; off >>= 1) n64 += __shfl_xor(n64, off);
        if (lane == 0u && n64) atomicAdd(segments, n64);
#ifdef RT_STATS
        for (int w = 1; w < 8; w++) {
            unsigned long long x = st.v[w];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (lane == 0u && x) atomicAdd(segments + w, x);
        }
        for (int w = 0; w < 7; w++)
            if (lane == 0u) atomicAdd(segments + 9 + w, st.cyc[w]);
#endif
    }
}

// The same for sphere lists beyond one LDS tile.  The workgroup streams the tiles of filter data in lock step, as
// render_body<true> does, one segment of its lanes' bounce rays per round (scan_tile, with a box around every tile and
// around every block of 32 spheres: most of the list is out of every lane's reach).  Camera rays need no tiles: a
// wavefront traces its unit's next 64 between the rounds, against the unit's sphere list (unit_list) -- or, for a unit
// with more spheres in reach than the list holds, against the launch's camera-ray terms in global memory.
//   LDS: [geo S float4][boxes 2 x S / 32 + 2 float4], S = RT_Q_STREAM_TILE, then per wavefront as in render_queue_body.
template <bool FOLD, class SC>
RT_DEV void render_queue_stream_body(const SC& sc, uint32_t* __restrict__ rgba, float4* __restrict__ linear,
                                     unsigned long long* __restrict__ segments, uint32_t* __restrict__ work,
                                     float4* __restrict__ fold_all,
                                     const RtCamera& cam, const RtLaunch& L) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: kept in a scalar register
    const uint64_t seed = ((uint64_t)L.seed_hi << 32) | L.seed_lo;
    const uint32_t flags = L.flags;
    const uint32_t P = L.unit_w * L.unit_h;
    const uint32_t n_samp = L.max_depth > 0u ? P * L.samples_per_pixel : 0u;   // per unit
    const uint32_t S = RT_Q_STREAM_TILE;
    const V3 T = ld3(L.translate);

    float4* geo = rt_lds;
    float4* box = geo + S;                                        // the boxes of the tile's blocks (scan_tile), 2 x S / 32
    const uint32_t box_f4 = L.scan_blocks ? 2u * (S / 32u) + 2u : 0u;   // ... and the tile's own box behind them
    // ... and two 64-bit words in which the wavefronts vote on the tiles of a round (below)
    unsigned long long* vote = reinterpret_cast<unsigned long long*>(box + box_f4);
    char* mine = reinterpret_cast<char*>(box + box_f4 + (box_f4 ? 1u : 0u)) + wave * (RT_Q_CAND_CAP * 128u + 48u * P + RT_Q_WORDS * 256u + RT_UL_BYTES(L.ulist_cap));
    Cand cand;
    cand.col = reinterpret_cast<uint16_t*>(mine) + lane;
    cand.n = 0u;
    cand.cap = RT_Q_CAND_CAP;
    unsigned long long* accum = reinterpret_cast<unsigned long long*>(mine + RT_Q_CAND_CAP * 128u);
    uint32_t* q = reinterpret_cast<uint32_t*>(mine + RT_Q_CAND_CAP * 128u + 48u * P);
    // the list of the unit whose camera rays are being traced (unit_list, from the camera-ray terms in global memory):
    // a camera-ray batch needs no sweep over the tiles and is traced between the workgroup's rounds.  A unit without a
    // list (more spheres in reach than the list holds) scans the launch's camera-ray terms in global memory instead.
    UnitList ul;
    ul.terms = reinterpret_cast<float4*>(mine + RT_Q_CAND_CAP * 128u + 48u * P + RT_Q_WORDS * 256u);
    ul.idx = reinterpret_cast<uint16_t*>(ul.terms + L.ulist_cap);
    uint32_t ul_k = RT_UL_NONE;
    const Tile whole = {geo, sc.exact(), nullptr, orig_of(L, sc.mat())};                           // exact-test data of the whole list, by sphere index
    for (uint32_t k = lane; k < 6u * P; k += 64u) accum[k] = 0ull;
    float4* fold = FOLD ? fold_all + (size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * (2u * P * (size_t)L.samples_per_pixel) : nullptr;
    const float4* fold1 = FOLD ? fold + (size_t)P * L.samples_per_pixel : nullptr;

    if (box_f4 && threadIdx.x == 0u) { vote[0] = 0ull; vote[1] = 0ull; }   // (the first round's barrier comes before the first vote)
    uint32_t round = 0u;
    Stats st;
    Slot u0, u1;
    u0.open = u1.open = false;
    u0.left = u1.left = 0u;
    u0.x0 = u0.ly0 = u0.y0 = u1.x0 = u1.ly0 = u1.y0 = 0u;
    uint32_t cur = 1u, next_samp = n_samp;
    bool drained = false;
    uint32_t q_head = 0u, q_count = 0u;
    bool need = true, active = false;
    Path p;
    p.o = v3(0.0f, 0.0f, 0.0f); p.d = v3(0.0f, 0.0f, 1.0f); p.thr = v3(1.0f, 1.0f, 1.0f); p.seg = 0u; p.nb = 0u;
    uint32_t pixel = 0u, s = 0u, pixslot = 0u, nseg = 0u;
    float band = 0.0f;

    for (;;) {
        // Between the workgroup's rounds: lanes without a path take queued rays; while some are still without one and
        // the queue is empty, the unit's next camera rays are traced on the spot (no tiles needed).
        unsigned long long want;
        for (;;) {
            if (u0.open && u0.left == 0u) { write_unit(L, u0, accum, lane, true, rgba, linear, fold); u0.open = false; }
            if (u1.open && u1.left == 0u) { write_unit(L, u1, accum + 3u * P, lane, true, rgba, linear, fold1); u1.open = false; }
            want = __ballot(need);
            if (want != 0ull && q_head < q_count) {
                const uint32_t j = q_head + __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                if (need && j < q_count) {
                    take_queued(sc, L, q, j, u0, u1, p, s, pixel, pixslot, band);
                    need = false;
                    active = true;
                }
                const uint32_t asked = (uint32_t)__popcll(want);
                q_head += asked < q_count - q_head ? asked : q_count - q_head;
                want = __ballot(need);
            }
            if (want == 0ull || q_head != q_count) break;
            if (next_samp >= n_samp) {
                if (drained || (cur ? u0.open : u1.open)) break;
                uint32_t unit = 0u;
                if (lane == 0u) unit = atomicAdd(work, 1u);
                unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
                if (unit >= L.n_units) { drained = true; break; }
                cur ^= 1u;
                next_samp = 0u;
                if (cur) u1 = open_unit(L, unit, n_samp); else u0 = open_unit(L, unit, n_samp);
                if (n_samp == 0u) continue;
                ul_k = L.ulist_cap ? uniform_u32(unit_list<true>(sc.geom, sc.exact(), sc.camterms(), cam, L, cur ? u1.x0 : u0.x0, cur ? u1.y0 : u0.y0, ul, lane)) : RT_UL_NONE;
            }
            uint32_t completed = 0u;
            primary_batch<true>(sc, whole, sc.camterms(), cam, L, seed, T, cur ? u1 : u0, cur, next_samp, n_samp, accum, fold, q, cand, ul, ul_k,
                                q_count, completed, nseg, st);
            q_head = 0u;
            next_samp = next_samp + 64u < n_samp ? next_samp + 64u : n_samp;
            if (cur) u1.left -= completed; else u0.left -= completed;
        }
        const bool more = active || q_head < q_count || !drained || next_samp < n_samp || u0.open || u1.open;
        if (!__syncthreads_or(more ? 1 : 0)) break;

        // The round: one segment of the lanes' bounce rays, the tiles streamed through LDS in lock step.
        Best b;
        b.t = L.t_max;
        b.idx = -1;
        bool finished = false;                                   // a bounce path of this lane ended in this round
        const float ra = vdot(p.d, p.d);
        // Which tiles does anybody need?  Every lane tests its ray against the boxes of the tiles (wave-uniform reads
        // from global memory), 64 tiles at a time; the wavefronts put their answers together in LDS, and a tile nobody's
        // ray reaches is neither staged nor waited for: one barrier for the vote instead of two per tile.  The two vote
        // words take turns: the one not in use is cleared after the barrier, a vote before it is used again.
        const V3 op = vsub(p.o, T);
        const RayTerms rterms = ray_terms(op, p.d, ra, L.scan_scale);
        BoxTerms bt;
        bt.inv = bt.n_lo = bt.n_hi = v3(0.0f, 0.0f, 0.0f);
        if (box_f4) bt = box_terms(op, rterms.uf, L.scan_scale);
        const float4* tboxes = sc.geom + L.n_padded + 2u * L.scan_blocks;
        const uint32_t n_tiles = (L.n_padded + RT_Q_STREAM_TILE - 1u) / RT_Q_STREAM_TILE;
        bool staged = false;
        for (uint32_t g0 = 0u; g0 < n_tiles; g0 += 64u) {
            const uint32_t gn = n_tiles - g0 < 64u ? n_tiles - g0 : 64u;
            unsigned long long needed = ~0ull;                      // tiles of this group some lane of the workgroup reaches
            unsigned long long mine_need = ~0ull;                   // ... some lane of this wavefront
            if (box_f4) {
                mine_need = 0ull;
                for (uint32_t t = 0u; t < gn; t++) {
                    float tn, tf;
                    box_span(bt, tboxes[2u * (g0 + t)], tboxes[2u * (g0 + t) + 1u], tn, tf);
                    if (__ballot(active && !(tf < tn)) != 0ull) mine_need |= 1ull << t;
                }
                if (lane == 0u && mine_need != 0ull) atomicOr(&vote[round & 1u], mine_need);
                __syncthreads();
                needed = vote[round & 1u];
                if (threadIdx.x == 0u) vote[(round & 1u) ^ 1u] = 0ull;
                round++;
            }
            for (uint32_t t = 0u; t < gn; t++) {
                if (!((needed >> t) & 1ull)) continue;               // (the same for the whole workgroup)
                const uint32_t first = (g0 + t) * RT_Q_STREAM_TILE;
                uint32_t count = L.n_padded - first;
                if (count > RT_Q_STREAM_TILE) count = RT_Q_STREAM_TILE;
                if (staged) __syncthreads();                         // previous tile fully scanned
                staged = true;
                for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) geo[i] = sc.geom[first + i];
                for (uint32_t i = threadIdx.x; i < box_f4; i += blockDim.x) {   // (2 float4 per 32 spheres, then the tile's own two)
                    if (i >= box_f4 - 2u) box[i] = tboxes[2u * (g0 + t) + (i - (box_f4 - 2u))];
                    else if (first / 16u + i < 2u * L.scan_blocks) box[i] = sc.geom[L.n_padded + first / 16u + i];
                }
                __syncthreads();
                const Tile tile = {geo, sc.exact(), box_f4 ? box : nullptr, orig_of(L, sc.mat()), box_f4 ? box + (box_f4 - 2u) : nullptr};   // exact-test data: the whole list's, in global memory
                // every lane of a wavefront that has a ray for this tile: see scan_tile
                if ((mine_need >> t) & 1ull)
                    scan_tile_stream(tile, count, (int)first, rterms, bt, p.o, p.d, ra, L.t_min, b, cand, st, active);
            }
        }
        drain(whole, 0, cand, p.o, p.d, ra, L.t_min, b, st);     // the segment's candidates, from all its tiles
        if (active) {
            nseg++;
            V3 color;
            if (shade_segment(sc, L, seed, b, pixel, s, band, p, color, st)) {
                add_sample(accum, P, pixslot, s, L.samples_per_pixel, color, fold);
                active = false;
                need = true;
                finished = true;
            }
        }
        const unsigned long long fin = __ballot(finished);
        if (fin != 0ull) {
            const uint32_t n1 = (uint32_t)__popcll(__ballot(finished && pixslot >= 3u * P));
            u1.left -= n1;
            u0.left -= (uint32_t)__popcll(fin) - n1;
        }
    }

    if (segments) {
        unsigned long long n64 = nseg;
        for (int off = 32; off > 0; off >>= 1) n64 += __shfl_xor(n64, off);
        if (lane == 0u && n64) atomicAdd(segments, n64);
#ifdef RT_STATS
        for (int w = 1; w < 8; w++) {
            unsigned long long x = st.v[w];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
            if (lane == 0u && x) atomicAdd(segments + w, x);
        }
        for (int w = 0; w < 7; w++)
            if (lane == 0u) atomicAdd(segments + 9 + w, st.cyc[w]);
#endif
    }
}

extern "C" __global__ void __launch_bounds__(1024)
rt_render(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
          float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
          RtCamera cam, RtLaunch L) {
    Scene sc = {geom, exact, shade, mat, camterms};
    RT_DBG_SET(segments);
    render_body<false>(sc, rgba, linear, segments, work, cam, L);
}

// Six waves per SIMD (77 VGPRs, nothing spilled) instead of
// five: the launcher's default of 512-thread workgroups lets LDS hold 24 wavefronts per compute unit; measured on one
// box against five waves and 256 threads: book scene 4K frame 950 -> 930 ms, deep-bounce hall 436 -> 420 ms.
extern "C" __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6)))
rt_render_queue(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
                float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
                RtCamera cam, RtLaunch L) {
    Scene sc = {geom, exact, shade, mat, camterms};
    RT_DBG_SET(segments);
    render_queue_body<false, false>(sc, rgba, linear, segments, work, fold, camterms, cam, L);
}

// rt_render_queue built with the blocks' bounding boxes in the bounce scan, for scenes whose sphere list the launcher
// sorted in space (32 spheres and more): on the 485-sphere book scene a bounce trip reaches 5 of the 16 blocks.
extern "C" __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6)))
rt_render_queue_boxes(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
                      float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
                      RtCamera cam, RtLaunch L) {
    Scene sc = {geom, exact, shade, mat, camterms};
    RT_DBG_SET(segments);
    render_queue_body<false, true>(sc, rgba, linear, segments, work, fold, camterms, cam, L);
}

extern "C" __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(RT_STREAM_WAVES)))
rt_render_queue_stream(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
                       float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
                       RtCamera cam, RtLaunch L) {
    SceneAt sc = {geom, L.n_padded};                            // (the other addresses: found from this one)
    RT_DBG_SET(segments);
    render_queue_stream_body<false>(sc, rgba, linear, segments, work, fold, cam, L);
}

// rt_render_queue with the pixel units held by the workgroup (render_pool_body).
extern "C" __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6)))
rt_render_pool(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
               float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
               RtCamera cam, RtLaunch L) {
    Scene sc = {geom, exact, shade, mat, camterms};
    RT_DBG_SET(segments);
    render_pool_body<false, true, false>(sc, rgba, linear, segments, work, nullptr, camterms, cam, L);
}

// rt_render_pool with the blocks' bounding boxes in the bounce scan (scenes sorted in space, as rt_render_queue_boxes).
extern "C" __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6)))
rt_render_pool_boxes(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
                     float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
                     RtCamera cam, RtLaunch L) {
    Scene sc = {geom, exact, shade, mat, camterms};
    RT_DBG_SET(segments);
    render_pool_body<false, true, true>(sc, rgba, linear, segments, work, nullptr, camterms, cam, L);
}

// rt_render_pool built without the pixel units' sphere lists, for the launches that have none (scenes of a handful of
// spheres, or no room in LDS): the list set-up's double-precision arithmetic costs the pooled kernel registers it is
// short of (two-sphere 1080p frame at 100 spp: 4.4 against 4.75 ms).
extern "C" __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6)))
rt_render_pool_plain(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
                     float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
                     RtCamera cam, RtLaunch L) {
    Scene sc = {geom, exact, shade, mat, camterms};
    RT_DBG_SET(segments);
    render_pool_body<false, false, false>(sc, rgba, linear, segments, work, nullptr, camterms, cam, L);
}

// The same two kernels in fold-order mode (RtParams.sample_chunk = 0): finished samples are parked in the
// wavefront's workspace and added in sample order in fp32 when their unit is complete -- the reference's
// summation order (ray.cu:123-129, lib.rs:228-231) at the speed of the exact-accumulation kernels.
extern "C" __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6)))
rt_render_queue_fold(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
                     float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
                     RtCamera cam, RtLaunch L) {
    Scene sc = {geom, exact, shade, mat, camterms};
    RT_DBG_SET(segments);
    render_queue_body<true, true>(sc, rgba, linear, segments, work, fold, camterms, cam, L);
}

// rt_render_pool in fold-order mode: finished samples go to the workgroup's part of the workspace.
extern "C" __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6)))
rt_render_pool_fold(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
                    float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
                    RtCamera cam, RtLaunch L) {
    Scene sc = {geom, exact, shade, mat, camterms};
    RT_DBG_SET(segments);
    render_pool_body<true, true, true>(sc, rgba, linear, segments, work, fold, camterms, cam, L);
}

extern "C" __global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(RT_STREAM_WAVES)))
rt_render_queue_stream_fold(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
                            float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
                            RtCamera cam, RtLaunch L) {
    SceneAt sc = {geom, L.n_padded};                            // (the other addresses: found from this one)
    RT_DBG_SET(segments);
    render_queue_stream_body<true>(sc, rgba, linear, segments, work, fold, cam, L);
}

extern "C" __global__ void __launch_bounds__(1024)
rt_render_stream(const float4* geom, const float4* exact, const float4* shade, const uint32_t* mat, uint32_t* rgba,
                 float4* linear, unsigned long long* segments, uint32_t* work, float4* fold, const float4* camterms,
                 RtCamera cam, RtLaunch L) {
    Scene sc = {geom, exact, shade, mat, camterms};
    RT_DBG_SET(segments);
    render_body<true>(sc, rgba, linear, segments, work, cam, L);
}

// ---------------------------------------------------------------- camera-ray terms
// Per sphere, the four numbers of scan_primary for this launch's camera: {A, B, C, sq}, all times rho.  ONE workgroup,
// a work-item per (padded) sphere and round, double precision; run on the launch stream before the camera-ray-batch
// kernels.  First pass: the unscaled terms and the largest M = |C| + (W+1)|A| + (H+1)|B| of the scene; second pass: the
// scale rho = 2^-k with rho * 2 M_max < 1 (scan_primary clamps rho * d.q to [0, 1]).  If some M is not finite, every
// sphere becomes an unconditional candidate (sq = -inf) and nothing is scaled.
//   exact[i] = {c, r*r} (r*r = -inf for padding)
extern "C" __global__ void __launch_bounds__(1024)
rt_camera_terms(const float4* __restrict__ exact, float4* __restrict__ camterms, RtCamera cam, RtLaunch L) {
    __shared__ unsigned long long m_max_bits;                    // bits of the largest M so far (M >= 0: ordered as integers; a NaN is larger than all)
    if (threadIdx.x == 0u) m_max_bits = 0ull;
    __syncthreads();
    const double o[3] = {cam.center[0], cam.center[1], cam.center[2]};
    const double du[3] = {cam.pixel_delta_u[0], cam.pixel_delta_u[1], cam.pixel_delta_u[2]};
    const double dv[3] = {cam.pixel_delta_v[0], cam.pixel_delta_v[1], cam.pixel_delta_v[2]};
    const double p0[3] = {(double)cam.pixel00_loc[0] - o[0], (double)cam.pixel00_loc[1] - o[1], (double)cam.pixel00_loc[2] - o[2]};
    const double W1 = (double)L.width + 1.0, H1 = (double)L.height + 1.0;   // |fx| <= W + 1, |fy| <= H + 1
    // F: distance from the camera centre to the image plane (a lower bound of |d|)
    const double n[3] = {du[1] * dv[2] - du[2] * dv[1], du[2] * dv[0] - du[0] * dv[2], du[0] * dv[1] - du[1] * dv[0]};
    const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const double F = fabs(p0[0] * n[0] + p0[1] * n[1] + p0[2] * n[2]) / nn;
    // eta: how far make_ray's rounded direction can be from P0 + fx du + fy dv.  Per component k its four roundings
    // (dv*fy, the fma, + pixel00, - o) are each below 2^-24 of a partial result, together
    // 2^-24 (|pixel00_k| + |P0_k| + 3 W1 |du_k| + 4 H1 |dv_k|)
    double eta = 0.0;
    for (int k = 0; k < 3; k++) {
        const double e = 0x1p-24 * (fabs((double)cam.pixel00_loc[k]) + fabs(p0[k]) + 3.0 * W1 * fabs(du[k]) + 4.0 * H1 * fabs(dv[k]));
        eta += e * e;
    }
    eta = sqrt(eta);

    // the longest camera ray direction of the frame (make_ray): |P0| + (W+1)|du| + (H+1)|dv|
    const double d_max = sqrt(p0[0] * p0[0] + p0[1] * p0[1] + p0[2] * p0[2]) + W1 * sqrt(du[0] * du[0] + du[1] * du[1] + du[2] * du[2]) +
                         H1 * sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
    // one sphere's unscaled terms; v = the right-hand side before rounding (-inf: unconditional candidate, +inf:
    // padding, -1: the half-space test alone, C already shifted)
    auto terms = [&](uint32_t i, double& A, double& B, double& C, double& v, double& M) {
        const float4 s = exact[i];
        const double q[3] = {(double)s.x - o[0], (double)s.y - o[1], (double)s.z - o[2]};
        A = du[0] * q[0] + du[1] * q[1] + du[2] * q[2];
        B = dv[0] * q[0] + dv[1] * q[1] + dv[2] * q[2];
        C = p0[0] * q[0] + p0[1] * q[1] + p0[2] * q[2];
        const double qq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
        M = fabs(C) + W1 * fabs(A) + H1 * fabs(B);
        const double s_abs = (4.0 * 0x1p-24 * M + 2.0 * sqrt(qq) * eta) / F + 0x1p-22 * sqrt(qq);
        const double ntm = qq * (1.0 - 0x1p-18) - (double)s.w * (1.0 + 0x1p-17);
        v = -HUGE_VAL;                                           // camera inside the (inflated) sphere, NaN, or no image plane
        if (ntm > 0.0 && F > 0.0 && s_abs == s_abs && s_abs < HUGE_VAL) {
            v = sqrt(ntm) * (1.0 - 0x1p-20) - s_abs;
            if (!(v > 0.0)) v = -HUGE_VAL;                       // nothing left of the bound
        }
        if (v == -HUGE_VAL) {
            // the exact test's own ncc for a camera ray (sphere_terms: the same single-precision operations, the same
            // for every ray of the frame).  ncc <= 0: not inside, so a hit needs h = d.q > 0.  The scan's dq differs from
            // the h the exact test computes by less than delta (roundings of A, B, C and the two fma: 4 * 2^-24 M; make_ray's
            // rounded direction against P0 + fx du + fy dv: |q| eta; the exact test's own dot product: 3 * 2^-24 |d| |q|);
            // with twice that added to C, dq <= 0 leaves no ray the exact test could accept.
            const float ocx = s.x - cam.center[0], ocy = s.y - cam.center[1], ocz = s.z - cam.center[2];
            const float ncc = fmaf(-ocz, ocz, fmaf(-ocy, ocy, fmaf(-ocx, ocx, s.w)));
            const double delta = 2.0 * (4.0 * 0x1p-24 * M + sqrt(qq) * eta + 3.0 * 0x1p-24 * d_max * sqrt(qq));
            if (ncc <= 0.0f && delta == delta && delta < HUGE_VAL) {
                C += delta;
                M += delta;
                v = -1.0;
            }
        }
        if (s.w == -INFINITY) { v = HUGE_VAL; A = B = C = M = 0.0; }   // padding: never a candidate
    };
    for (uint32_t i = threadIdx.x; i < L.n_padded; i += blockDim.x) {
        double A, B, C, v, M;
        terms(i, A, B, C, v, M);
        atomicMax(&m_max_bits, (unsigned long long)__double_as_longlong(M));
    }
    __syncthreads();
    const double m_max = __longlong_as_double((long long)m_max_bits);
    const bool bounded = m_max < HUGE_VAL;                       // false for inf and NaN
    double rho = 1.0;
    if (bounded && m_max > 0.0) {
        int k = 0;
        (void)frexp(2.0 * m_max, &k);                            // 2 M = f 2^k, f in [0.5, 1)
        rho = ldexp(1.0, -k);
    }
    for (uint32_t i = threadIdx.x; i < L.n_padded; i += blockDim.x) {
        double A, B, C, v, M;
        terms(i, A, B, C, v, M);
        if (!bounded && v != HUGE_VAL) v = -HUGE_VAL;
        float sq;
        if (v == -1.0) {
            sq = 0x1p-126f;                                      // half-space test: candidate iff rho (dq + delta) iLs does not vanish
        } else {
            v *= rho;                                            // infinities stay
            sq = (float)v;
            if (v > -HUGE_VAL && v < HUGE_VAL && ((double)sq > v || sq != sq))
                sq = __uint_as_float(__float_as_uint(sq) + (sq > 0.0f ? 0xffffffffu : 1u));   // round down
        }
        camterms[i] = make_float4((float)(A * rho), (float)(B * rho), (float)(C * rho), sq);
    }
}

// ---------------------------------------------------------------- primary-hit index map
// Closest sphere index (or -1) of the ray through each pixel centre (sub-pixel offset 0).
extern "C" __global__ void __launch_bounds__(1024)
rt_primary_hits(const float4* geom, const float4* exact, const uint32_t* mat, int32_t* hits, RtCamera cam, RtLaunch L) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t x, ly, y;
    const bool valid = map_pixel(L, wave_tile, lane, x, ly, y);
    V3 o, d;
    float fx_unused, fy_unused;
    make_ray(cam, x, y, 0.0f, 0.0f, o, d, fx_unused, fy_unused);
    const float a = vdot(d, d);
    Best b;
    b.t = L.t_max;
    b.idx = -1;
    Cand cand = cand_list(L);
    Stats st;
    for (uint32_t first = 0u; first < L.n_padded; first += RT_TILE_SPHERES) {
        uint32_t count = L.n_padded - first;
        if (count > RT_TILE_SPHERES) count = RT_TILE_SPHERES;
        if (first) __syncthreads();
        Tile tile = stage_tile(L, geom, exact, orig_of(L, mat), first, count);
        __syncthreads();
        if (valid) scan_tile<false>(tile, count, (int)first, ld3(L.translate), L.scan_scale, o, d, a, L.t_min, b, cand, st);
    }
    if (valid) hits[(size_t)ly * L.width + x] = b.idx < 0 ? -1 : L.sorted ? (int32_t)mat[L.n_padded + (uint32_t)b.idx] : b.idx;   // the caller's index
}

// ---------------------------------------------------------------- RNG debug views (ray.cu:114-118)
extern "C" __global__ void __launch_bounds__(1024)
rt_show_random(uint32_t* rgba, uint32_t n_pixels, uint32_t seed_lo, uint32_t seed_hi, uint32_t norm) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    Rng g = stream_init(((uint64_t)seed_hi << 32) | seed_lo, p, 0u, 0u);
    V3 c;
    if (norm) {
        Stats st;
        c = random_unit_vector(g, st);
    } else {
        c.x = rand01(g); c.y = rand01(g); c.z = rand01(g);
    }
    // to_pixel without gamma (ray.cu:81-89,116)
    float v[3] = {c.x * 255.0f, c.y * 255.0f, c.z * 255.0f};
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        float t = v[k];
        if (!(t == t)) { q[k] = 255u; continue; }
        t = t < 255.0f ? t : 255.0f;
        t = t > 0.0f ? t : 0.0f;
        q[k] = (uint32_t)t;
    }
    rgba[p] = q[0] | (q[1] << 8) | (q[2] << 16) | (255u << 24);
}

// ---------------------------------------------------------------- multi-GPU frame assembly
// `gathered` holds, rank after rank, each rank's compact buffer of interleaved row tiles (rank r owns tiles
// r, r + world, ...; rows_per_rank rows each, padded to the largest rank); writes the frame in image order.
// One RGBA8 pixel per work-item, grid-stride.
extern "C" __global__ void __launch_bounds__(256)
rt_assemble(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ frame, uint32_t width, uint32_t height,
            uint32_t tile_rows, uint32_t world, uint32_t rows_per_rank) {
    const size_t n = (size_t)width * height;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const uint32_t y = (uint32_t)(idx / width), x = (uint32_t)(idx - (size_t)y * width);
        const uint32_t t = y / tile_rows, r = t % world, k = t / world;
        const uint32_t local_row = k * tile_rows + (y - t * tile_rows);
        frame[idx] = gathered[((size_t)r * rows_per_rank + local_row) * width + x];
    }
}
