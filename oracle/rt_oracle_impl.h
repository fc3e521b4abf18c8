/*
 * rt_oracle_impl.h -- body of the CPU restatement, included twice by rt_oracle.c:
 *   OR_F32 = 0: REAL = double, MADD(a,b,c) = a*b + c with two roundings   -> or64_*
 *   OR_F32 = 1: REAL = float,  MADD(a,b,c) = fmaf(a,b,c)                  -> or32_*
 * Built with -ffp-contract=off so the compiler adds no fusion of its own.
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * Citations: "lib.rs" = cpu-implementation/src/lib.rs; "library.cu", "ray.cu",
 * "floatN_helper.cu" = gpu-implementation/src/cuda/...; "ray_tracing.rs" =
 * gpu-implementation/src/ray_tracing.rs.
 */

#if OR_F32
#  define REAL float
#  define NAME(x) or32_##x
#  define CAM OrCameraF
#  define R_SQRT(x) sqrtf(x)
#  define R_FABS(x) fabsf(x)
#  define R_FMIN(a, b) fminf(a, b)
#  define R_INF INFINITY
static inline float or32_madd(float a, float b, float c) { return fmaf(a, b, c); }
#else
#  define REAL double
#  define NAME(x) or64_##x
#  define CAM OrCameraD
#  define R_SQRT(x) sqrt(x)
#  define R_FABS(x) fabs(x)
#  define R_FMIN(a, b) fmin(a, b)
#  define R_INF ((double)INFINITY)
static inline double or64_madd(double a, double b, double c) { return a * b + c; }
#endif
#define MADD(a, b, c) NAME(madd)((a), (b), (c))
#define NMADD(a, b, c) NAME(madd)(-(a), (b), (c)) /* c - a*b */
#define R(x) ((REAL)(x))

typedef struct { REAL x, y, z; } NAME(V3);
#define V3 NAME(V3)

static inline V3 NAME(v3)(REAL x, REAL y, REAL z) { V3 v = {x, y, z}; return v; }
static inline V3 NAME(sub)(V3 a, V3 b) { return NAME(v3)(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 NAME(add)(V3 a, V3 b) { return NAME(v3)(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 NAME(scale)(V3 a, REAL s) { return NAME(v3)(a.x * s, a.y * s, a.z * s); }
/* a.x*b.x + a.y*b.y + a.z*b.z  (floatN_helper.cu:109-111; cgmath dot, lib.rs:43,99-101) */
static inline REAL NAME(dot)(V3 a, V3 b) { return MADD(a.z, b.z, MADD(a.y, b.y, a.x * b.x)); }
/* v * (1/|v|): cgmath `normalize` (lib.rs:44,203), this form reproduces image.ppm to the last bit;
 * stands in for CUDA's rnorm3df in floatN_helper.cu:120-122. */
static inline V3 NAME(unit)(V3 v) {
    REAL inv = R(1) / R_SQRT(NAME(dot)(v, v));
    return NAME(scale)(v, inv);
}
static inline V3 NAME(ld3f)(const float *p) { return NAME(v3)(R(p[0]), R(p[1]), R(p[2])); }

typedef struct { uint64_t state, inc; } NAME(Rng);
static inline REAL NAME(rand01)(NAME(Rng) *g) { return R(or_u01(or_pcg32_next(&g->state, g->inc))); }

/* random_unit_vector: rejection in (-1,1)^3, accept p.p <= 1, normalise (lib.rs:31-47), with the
 * GPU path's bound of 20 tries and its (1,0,0) fallback (library.cu:25-31).  One try takes ONE generator
 * output a and cuts three coordinates out of its 32 bits: x = a >> 21 and y = (a >> 10) & 0x7ff (11 bits each,
 * value (k + 1/2) 2^-10 - 1), z = a & 0x3ff (10 bits, (k + 1/2) 2^-9 - 1): cell centres of a regular grid over the
 * cube, symmetric about the origin, never the zero vector; every value is exact in single precision. */
static inline V3 NAME(random_unit_vector)(NAME(Rng) *g) {
    for (int i = 0; i < 20; i++) {
        uint32_t a = or_pcg32_next(&g->state, g->inc);
        REAL x = MADD(R(a >> 21), R(0x1p-10), R(-1 + 0x1p-11));
        REAL y = MADD(R((a >> 10) & 0x7ffu), R(0x1p-10), R(-1 + 0x1p-11));
        REAL z = MADD(R(a & 0x3ffu), R(0x1p-9), R(-1 + 0x1p-10));
        V3 p = NAME(v3)(x, y, z);
        if (NAME(dot)(p, p) <= R(1)) return NAME(unit)(p);
    }
    return NAME(v3)(R(1), R(0), R(0));
}

typedef struct {
    V3 c;        /* center */
    REAL r, rr;  /* radius, radius*radius */
    uint32_t mat;
    V3 albedo;
    REAL param;
} NAME(Sph);

static NAME(Sph) *NAME(convert_scene)(const RtSphere *s, uint32_t n) {
    NAME(Sph) *o = (NAME(Sph) *)malloc(sizeof(NAME(Sph)) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) {
        o[i].c = NAME(ld3f)(s[i].center);
        o[i].r = R(s[i].radius);
        o[i].rr = o[i].r * o[i].r; /* radius*radius (lib.rs:101, library.cu:203) */
        o[i].mat = s[i].material;
        o[i].albedo = NAME(ld3f)(s[i].albedo);
        float p = s[i].param;
        if (s[i].material == RT_MAT_REFLECT) p = p < 1.0f ? p : 1.0f; /* fuzz clamp, library.cu:144 */
        o[i].param = R(p);
    }
    return o;
}

typedef struct { int32_t idx; REAL t; V3 p, n; int front; } NAME(Hit);

/* Closest hit over the list: Sphere::hit (lib.rs:96-125, library.cu:198-227) folded with a shrinking
 * upper bound so that a later sphere replaces the record only if strictly closer
 * (lib.rs:127-135, ray.cu:1-18).  `a` = d.d is the same for every sphere and computed once. */
static inline NAME(Hit) NAME(closest_hit)(const NAME(Sph) *w, uint32_t n, V3 o, V3 d, REAL t_min, REAL t_max, uint32_t flags) {
    NAME(Hit) best;
    best.idx = -1; best.t = t_max; best.front = 0;
    best.p = NAME(v3)(0, 0, 0); best.n = best.p;
    REAL a = NAME(dot)(d, d);
    REAL best_h = 0, best_s = 0; (void)best_h; (void)best_s;
    for (uint32_t i = 0; i < n; i++) {
        V3 oc = NAME(sub)(w[i].c, o);
        REAL h = NAME(dot)(d, oc);
#if OR_F32
        /* -(oc.oc - r*r) accumulated from r*r downwards; disc = h*h - a*c */
        REAL ncc = NMADD(oc.z, oc.z, NMADD(oc.y, oc.y, NMADD(oc.x, oc.x, w[i].rr)));
        REAL disc = MADD(h, h, a * ncc);
#else
        REAL c = NAME(dot)(oc, oc) - w[i].rr;
        REAL disc = h * h - a * c;
#endif
        if (disc < R(0)) continue;
        REAL s = R_SQRT(disc);
        REAL root_a = (h - s) / a;
        REAL root_b = (h + s) / a;
        REAL root;
        if (t_min < root_a && root_a < best.t) root = root_a;      /* Bound::contains lib.rs:255-257, floatN_helper.cu:113-115 */
        else if (t_min < root_b && root_b < best.t) root = root_b;
        else continue;
        best.idx = (int32_t)i;
        best.t = root;
    }
    if (best.idx >= 0) {
        const NAME(Sph) *sp = &w[best.idx];
        /* ray.at(root) = orig + dir*root (lib.rs:20-22, library.cu:98-100) */
        best.p = NAME(v3)(MADD(d.x, best.t, o.x), MADD(d.y, best.t, o.y), MADD(d.z, best.t, o.z));
        V3 pc = NAME(sub)(best.p, sp->c);
        V3 nrm = NAME(v3)(pc.x / sp->r, pc.y / sp->r, pc.z / sp->r); /* (point - center)/radius */
        REAL nd = NAME(dot)(nrm, d);
        int flip;
        if (flags & RT_FLAG_FLIP_SIGNBIT) flip = !signbit(nd);   /* lib.rs:49-58 with is_negative=false */
        else flip = nd > R(0);                                    /* floatN_helper.cu:129-136 */
        if (flip) nrm = NAME(v3)(-nrm.x, -nrm.y, -nrm.z);
        best.n = nrm;
        best.front = !flip;
    }
    return best;
}

/* Primary ray for pixel (i,j) with sub-pixel offset (si,sj):
 * pixel00 + du*(i+si) + dv*(j+sj) - center  (ray.cu:62-77); with du=(dx,0,0), dv=(0,dy,0) this is
 * lib.rs:215-217 `(offset + pixel) * pixel_delta + pixel00_loc - center` value for value. */
static inline void NAME(make_ray)(const CAM *cam, uint32_t i, uint32_t j, REAL si, REAL sj, V3 *o, V3 *d) {
    REAL fx = R(i) + si, fy = R(j) + sj;
    V3 pc;
    pc.x = cam->pixel00_loc[0] + MADD(cam->pixel_delta_u[0], fx, cam->pixel_delta_v[0] * fy);
    pc.y = cam->pixel00_loc[1] + MADD(cam->pixel_delta_u[1], fx, cam->pixel_delta_v[1] * fy);
    pc.z = cam->pixel00_loc[2] + MADD(cam->pixel_delta_u[2], fx, cam->pixel_delta_v[2] * fy);
    *o = NAME(v3)(cam->center[0], cam->center[1], cam->center[2]);
    *d = NAME(sub)(pc, *o);
}

void NAME(pixel_center_ray)(const CAM *cam, uint32_t x, uint32_t y, REAL orig[3], REAL dir[3]) {
    V3 o, d;
    NAME(make_ray)(cam, x, y, R(0), R(0), &o, &d);
    orig[0] = o.x; orig[1] = o.y; orig[2] = o.z;
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

/* (1-a)*white + a*(0.5,0.7,1.0) with a = blend*(unit(d).y + 1)  (lib.rs:203-205; ray.cu:43-45 has blend 0.5
 * written as (y+1)*0.5, the same product). */
static inline V3 NAME(sky)(V3 d, REAL blend) {
    V3 u = NAME(unit)(d);
    REAL a = blend * (u.y + R(1));
    REAL om = R(1) - a;
    return NAME(v3)(MADD(a, R(0.5), om), MADD(a, R(0.7), om), MADD(a, R(1.0), om));
}

void NAME(miss_color)(const REAL dir[3], REAL blend, REAL rgb[3]) {
    V3 c = NAME(sky)(NAME(v3)(dir[0], dir[1], dir[2]), blend);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* Material::scatter (library.cu:131-149,175-183) plus the book's dielectric.  Returns the new direction;
 * *atten receives the attenuation.  The segment's stream (seed, pixel, sample, k) is started only by a
 * material that draws from it: Lambertian and metal with fuzz > 0 draw the unit-vector tries, glass draws one
 * uniform (reflect or refract), a mirror (fuzz 0) draws nothing. */
static inline V3 NAME(scatter)(const NAME(Sph) *sp, V3 d, const NAME(Hit) *h, uint64_t seed, uint32_t pixel,
                               uint32_t sample, uint32_t k, uint32_t flags, V3 *atten) {
    const int band = (flags & RT_FLAG_BAND_ALBEDO) != 0;
    const int glass = !band && sp->mat == RT_MAT_DIELECTRIC;
    const int wants_unit = band || sp->mat == RT_MAT_DIFFUSE || (sp->mat == RT_MAT_REFLECT && sp->param != R(0));
    REAL u_choice = R(0);
    V3 ru = NAME(v3)(R(0), R(0), R(0));
    if (glass || wants_unit) {
        NAME(Rng) g;
        or_stream_init(&g.state, &g.inc, seed, pixel, sample, k);
        if (glass) u_choice = NAME(rand01)(&g);
        else ru = NAME(random_unit_vector)(&g);
    }
    V3 n = h->n;
    *atten = sp->albedo;
    if ((flags & RT_FLAG_BAND_ALBEDO) || sp->mat == RT_MAT_DIFFUSE) {
        V3 dir = NAME(add)(n, ru); /* record.normal + random unit (lib.rs:194, library.cu:132) */
        if (!(flags & RT_FLAG_BAND_ALBEDO)) {
            REAL s = R(1e-8f); /* float3_near_zero_mag, floatN_helper.cu:116-119 */
            if (R_FABS(dir.x) < s && R_FABS(dir.y) < s && R_FABS(dir.z) < s) dir = n;
        }
        return dir;
    }
    if (sp->mat == RT_MAT_REFLECT) {
        /* a - b*2*dot(a,b) (floatN_helper.cu:124-127); normalize + fuzz*random (library.cu:146-147) */
        REAL k = R(2) * NAME(dot)(d, n);
        V3 refl = NAME(v3)(NMADD(n.x, k, d.x), NMADD(n.y, k, d.y), NMADD(n.z, k, d.z));
        V3 ur = NAME(unit)(refl);
        /* fuzz * random (library.cu:147) adds nothing when fuzz is 0: the mirror direction itself */
        if (!wants_unit) return ur;
        return NAME(v3)(MADD(ru.x, sp->param, ur.x), MADD(ru.y, sp->param, ur.y), MADD(ru.z, sp->param, ur.z));
    }
    /* Dielectric: not in the reference; "Ray Tracing in One Weekend" section 11 (Snell + Schlick). */
    REAL ri = h->front ? R(1) / sp->param : sp->param;
    V3 ud = NAME(unit)(d);
    REAL udn = NAME(dot)(ud, n);
    REAL cos_t = R_FMIN(-udn, R(1));
    REAL sin_t = R_SQRT(NMADD(cos_t, cos_t, R(1)));
    int cannot = ri * sin_t > R(1);
    REAL r0 = (R(1) - ri) / (R(1) + ri);
    r0 = r0 * r0;
    REAL x = R(1) - cos_t;
    REAL x2 = x * x;
    REAL x5 = x2 * x2 * x;
    REAL prob = MADD(R(1) - r0, x5, r0);
    if (cannot || prob > u_choice) {
        REAL k = R(2) * udn;
        return NAME(v3)(NMADD(n.x, k, ud.x), NMADD(n.y, k, ud.y), NMADD(n.z, k, ud.z));
    }
    V3 perp = NAME(v3)(ri * MADD(cos_t, n.x, ud.x), ri * MADD(cos_t, n.y, ud.y), ri * MADD(cos_t, n.z, ud.z));
    REAL par = -R_SQRT(R_FABS(R(1) - NAME(dot)(perp, perp)));
    return NAME(v3)(MADD(par, n.x, perp.x), MADD(par, n.y, perp.y), MADD(par, n.z, perp.z));
}

/* One scatter event by itself, for known-answer tests: sphere material (tag, fuzz | ior), incoming direction,
 * surface normal facing the ray, front-face flag, and the stream key (seed, pixel, sample, k). */
void NAME(scatter_dir)(uint32_t material, REAL param, const REAL d[3], const REAL n[3], int front, uint64_t seed,
                       uint32_t pixel, uint32_t sample, uint32_t k, uint32_t flags, REAL out[3]) {
    NAME(Sph) sp;
    sp.c = NAME(v3)(R(0), R(0), R(0)); sp.r = R(1); sp.rr = R(1);
    sp.mat = material;
    sp.albedo = NAME(v3)(R(1), R(1), R(1));
    float pf = (float)param;
    if (material == RT_MAT_REFLECT) pf = pf < 1.0f ? pf : 1.0f; /* fuzz clamp, library.cu:144 */
    sp.param = R(pf);
    NAME(Hit) h;
    h.idx = 0; h.t = R(1); h.p = NAME(v3)(R(0), R(0), R(0));
    h.n = NAME(v3)(n[0], n[1], n[2]);
    h.front = front;
    V3 atten;
    V3 o = NAME(scatter)(&sp, NAME(v3)(d[0], d[1], d[2]), &h, seed, pixel, sample, k, flags, &atten);
    out[0] = o.x; out[1] = o.y; out[2] = o.z;
}

/* The unit vector a stream's rejection loop yields (lib.rs:31-47), by itself. */
void NAME(unit_vector)(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k, REAL out[3]) {
    NAME(Rng) g;
    or_stream_init(&g.state, &g.inc, seed, pixel, sample, k);
    V3 v = NAME(random_unit_vector)(&g);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}

/* ray_color (lib.rs:184-206 recursive, ray.cu:22-49 iterative): radiance of one path, counting segments. */
static inline V3 NAME(ray_color)(const NAME(Sph) *w, uint32_t n, V3 o, V3 d, const RtParams *prm,
                                 uint32_t pixel, uint32_t sample, REAL band, uint64_t *nseg, uint64_t *sig) {
    const uint32_t flags = prm->flags;
    const REAL t_min = R(prm->t_min), t_max = R(prm->t_max);
    V3 thr = NAME(v3)(R(1), R(1), R(1));
    uint32_t nb = 0;
    for (uint32_t seg = 0;; seg++) {
        if (seg == prm->max_depth) return NAME(v3)(R(0), R(0), R(0)); /* lib.rs:190-192, ray.cu:30-32 */
        (*nseg)++;
        NAME(Hit) h = NAME(closest_hit)(w, n, o, d, t_min, t_max, flags);
        if (sig) *sig = (*sig ^ (uint64_t)(uint32_t)(h.idx + 2)) * 0x100000001B3ULL; /* FNV-1a over the hit indices */
        if (h.idx >= 0) {
            V3 atten;
            V3 nd = NAME(scatter)(&w[h.idx], d, &h, prm->seed, pixel, sample, seg + 1, flags, &atten);
            o = h.p;
            d = nd;
            if (flags & RT_FLAG_BAND_ALBEDO) nb++;
            else thr = NAME(v3)(thr.x * atten.x, thr.y * atten.y, thr.z * atten.z); /* ray.cu:37 */
            continue;
        }
        if (flags & RT_FLAG_BAND_ALBEDO) {
            /* reflectance * ray_color(...) unwound from the innermost call outwards (lib.rs:195-205) */
            V3 c = NAME(sky)(d, band);
            for (uint32_t k = 0; k < nb; k++) c = NAME(v3)(band * c.x, band * c.y, band * c.z);
            return c;
        }
        V3 s = NAME(sky)(d, R(0.5));
        return NAME(v3)(thr.x * s.x, thr.y * s.y, thr.z * s.z); /* ray.cu:45 */
    }
}

/* Run sum -> 2^-32 fixed point: floor part and the truncated fraction; values that are not finite and below
 * 2^30 in magnitude count as 0 (RtParams.sample_chunk). */
static inline int64_t NAME(to_fixed)(REAL x) {
    if (!(R_FABS(x) < R(1073741824.0))) return 0;
#if OR_F32
    REAL hi = floorf(x);
#else
    REAL hi = floor(x);
#endif
    REAL frac = x - hi; /* in [0,1], exact unless x is a tiny negative number: then hi = -1 and x + 1 rounds to 1 */
    if (frac >= R(1.0)) { hi += R(1.0); frac = R(0.0); }
    uint32_t lo = (uint32_t)(frac * R(4294967296.0)); /* truncation */
    return (int64_t)((uint64_t)(int64_t)(int32_t)hi << 32) + (int64_t)lo;
}

static inline REAL NAME(fixed_mean)(int64_t sum, uint32_t spp) {
    return (REAL)((double)sum * 0x1p-32 / (double)spp);
}

static inline uint8_t NAME(quant)(REAL lin, uint32_t flags) {
    REAL g = R_SQRT(lin); /* gamma 2 (lib.rs:236, ray.cu:129) */
    REAL v = g * R(255);
    if (flags & RT_FLAG_QUANT_ROUND) { /* `(x.sqrt()*255.0).round() as u8`: half away from zero, saturating, NaN -> 0 */
        if (!(v == v)) return 0;
#if OR_F32
        REAL r = roundf(v);
#else
        REAL r = round(v);
#endif
        if (r <= R(0)) return 0;
        if (r >= R(255)) return 255;
        return (uint8_t)r;
    }
    /* (unsigned char) fmaxf(0, fminf(x*255, 255)) (ray.cu:81-89): fminf drops a NaN operand */
    if (!(v == v)) return 255;
    REAL c = v < R(255) ? v : R(255);
    c = c > R(0) ? c : R(0);
    return (uint8_t)c;
}

typedef struct {
    const CAM *cam;
    const NAME(Sph) *w;
    uint32_t n;
    const RtParams *prm;
    uint32_t row0, rows;
    uint8_t *rgba;
    REAL *linear;
    uint64_t segments;
    uint64_t *sig; /* optional, per pixel: hash of the sphere indices its paths hit, in order (-1 = sky) */
    /* primary-hit mode */
    int hits_mode;
    REAL t_min, t_max;
    uint32_t flags;
    int32_t *idx;
    REAL *t, *point, *normal;
    volatile uint32_t *next_row;
} NAME(Job);

static void NAME(render_row)(NAME(Job) *jb, uint32_t lr) {
    const CAM *cam = jb->cam;
    const uint32_t W = cam->image_width;
    const uint32_t y = jb->row0 + lr;
    if (jb->hits_mode) {
        for (uint32_t x = 0; x < W; x++) {
            V3 o, d;
            NAME(make_ray)(cam, x, y, R(0), R(0), &o, &d);
            NAME(Hit) h = NAME(closest_hit)(jb->w, jb->n, o, d, jb->t_min, jb->t_max, jb->flags);
            size_t k = (size_t)lr * W + x;
            jb->idx[k] = h.idx;
            if (jb->t) jb->t[k] = h.idx >= 0 ? h.t : R(0);
            if (jb->point) { jb->point[3 * k] = h.p.x; jb->point[3 * k + 1] = h.p.y; jb->point[3 * k + 2] = h.p.z; }
            if (jb->normal) { jb->normal[3 * k] = h.n.x; jb->normal[3 * k + 1] = h.n.y; jb->normal[3 * k + 2] = h.n.z; }
        }
        return;
    }
    const RtParams *prm = jb->prm;
    const uint32_t spp = prm->samples_per_pixel;
    uint64_t nseg = 0;
    for (uint32_t x = 0; x < W; x++) {
        const uint32_t pixel = y * W + x;
        REAL band = R(0);
        if (prm->flags & RT_FLAG_BAND_ALBEDO) {
            uint32_t gi = W / 5; /* lib.rs:225-226 */
            band = R(0.1) + R(0.2) * R(x / (gi ? gi : 1));
        }
        V3 total = NAME(v3)(R(0), R(0), R(0));
        const uint32_t chunk = (prm->sample_chunk && prm->sample_chunk < spp) ? prm->sample_chunk : 0;
        int64_t fx = 0, fy = 0, fz = 0;
        uint64_t sig = 0xcbf29ce484222325ULL;
        for (uint32_t s = 0; s < spp; s++) {
            REAL si, sj;
            if (prm->flags & RT_FLAG_SAMPLER_GRID) { /* ray.cu:59-60 */
                uint32_t n2 = prm->sample2_per_pixel;
                si = R(s % n2) / R(n2);
                sj = R(s / n2) / R(n2);
            } else { /* lib.rs:209-214: two draws in [-0.5, 0.5) */
                NAME(Rng) g;
                or_stream_init(&g.state, &g.inc, prm->seed, pixel, s, 0);
                si = NAME(rand01)(&g) - R(0.5);
                sj = NAME(rand01)(&g) - R(0.5);
            }
            V3 o, d;
            NAME(make_ray)(cam, x, y, si, sj, &o, &d);
            V3 c = NAME(ray_color)(jb->w, jb->n, o, d, prm, pixel, s, band, &nseg, jb->sig ? &sig : NULL);
            total = NAME(add)(total, c); /* lib.rs:228-231, ray.cu:126 */
            if (chunk && ((s + 1) % chunk == 0 || s + 1 == spp)) { /* end of a run: exact accumulation */
                fx += NAME(to_fixed)(total.x); fy += NAME(to_fixed)(total.y); fz += NAME(to_fixed)(total.z);
                total = NAME(v3)(R(0), R(0), R(0));
            }
        }
        V3 mean = NAME(v3)(total.x / R(spp), total.y / R(spp), total.z / R(spp));
        if (chunk) mean = NAME(v3)(NAME(fixed_mean)(fx, spp), NAME(fixed_mean)(fy, spp), NAME(fixed_mean)(fz, spp));
        size_t k = (size_t)lr * W + x;
        if (jb->sig) jb->sig[k] = sig;
        if (jb->rgba) {
            jb->rgba[4 * k + 0] = NAME(quant)(mean.x, prm->flags);
            jb->rgba[4 * k + 1] = NAME(quant)(mean.y, prm->flags);
            jb->rgba[4 * k + 2] = NAME(quant)(mean.z, prm->flags);
            jb->rgba[4 * k + 3] = 255;
        }
        if (jb->linear) {
            jb->linear[4 * k + 0] = mean.x; jb->linear[4 * k + 1] = mean.y;
            jb->linear[4 * k + 2] = mean.z; jb->linear[4 * k + 3] = R(1);
        }
    }
    __atomic_fetch_add(&jb->segments, nseg, __ATOMIC_RELAXED);
}

static void *NAME(worker)(void *arg) {
    NAME(Job) *jb = (NAME(Job) *)arg;
    for (;;) {
        uint32_t lr = __atomic_fetch_add(jb->next_row, 1u, __ATOMIC_RELAXED);
        if (lr >= jb->rows) break;
        NAME(render_row)(jb, lr);
    }
    return NULL;
}

/* Rows are handed to `threads` workers one at a time, as rayon hands out pixels in lib.rs:222. */
static void NAME(run_job)(NAME(Job) *jb, int threads) {
    uint32_t next = 0;
    jb->next_row = &next;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (threads == 1) { NAME(worker)(jb); return; }
    pthread_t th[256];
    int started = 0;
    for (int i = 0; i < threads; i++)
        if (pthread_create(&th[started], NULL, NAME(worker), jb) == 0) started++;
    if (started == 0) NAME(worker)(jb);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
}

void NAME(primary_hits)(const CAM *cam, const RtSphere *spheres, uint32_t n, REAL t_min, REAL t_max,
                        uint32_t flags, uint32_t row0, uint32_t rows, int threads,
                        int32_t *idx, REAL *t, REAL *point, REAL *normal) {
    NAME(Job) jb;
    memset(&jb, 0, sizeof jb);
    NAME(Sph) *w = NAME(convert_scene)(spheres, n);
    jb.cam = cam; jb.w = w; jb.n = n; jb.row0 = row0; jb.rows = rows;
    jb.hits_mode = 1; jb.t_min = t_min; jb.t_max = t_max; jb.flags = flags;
    jb.idx = idx; jb.t = t; jb.point = point; jb.normal = normal;
    NAME(run_job)(&jb, threads);
    free(w);
}

void NAME(render_sig)(const CAM *cam, const RtSphere *spheres, uint32_t n, const RtParams *params,
                      uint32_t row0, uint32_t rows, int threads,
                      uint8_t *rgba, REAL *linear, uint64_t *segments, uint64_t *path_sig) {
    NAME(Job) jb;
    memset(&jb, 0, sizeof jb);
    NAME(Sph) *w = NAME(convert_scene)(spheres, n);
    jb.cam = cam; jb.w = w; jb.n = n; jb.prm = params; jb.row0 = row0; jb.rows = rows;
    jb.rgba = rgba; jb.linear = linear; jb.sig = path_sig;
    NAME(run_job)(&jb, threads);
    if (segments) *segments = jb.segments;
    free(w);
}

void NAME(render)(const CAM *cam, const RtSphere *spheres, uint32_t n, const RtParams *params,
                  uint32_t row0, uint32_t rows, int threads,
                  uint8_t *rgba, REAL *linear, uint64_t *segments) {
    NAME(render_sig)(cam, spheres, n, params, row0, rows, threads, rgba, linear, segments, NULL);
}

#undef REAL
#undef NAME
#undef CAM
#undef R_SQRT
#undef R_FABS
#undef R_FMIN
#undef R_INF
#undef MADD
#undef NMADD
#undef R
#undef V3
