"""The scan's conservative filter must never drop a sphere the exact test can accept.

Kernel (rt_kernels.hip, scan_tile), with T the scene translation, c' = fl(c - T), o' = fl(o - T):
  t = fma(w.z,c'z, fma(w.y,c'y, fma(w.x,c'x, k0)))     w = 2 o',  k0 = -(o'.o' (1-2^-18))
  g = fma(uf.z,c'z, fma(uf.y,c'y, fma(uf.x,c'x, e)))   uf = d rsq(d.d) (1+2^-19),  e = -uf.o'
                                                       rsq = the hardware's 1/sqrt, accurate to 1 ulp: modelled here as the
                                                       correctly rounded value and its two neighbours (all three are checked)
  key = fma(g, g, t) + m                               m = r*r(1+2^-17) - |c'|^2 (1-2^-18), rounded up;
  candidate iff sign(key) clear.
Exact test (checker order):  disc = fma(h, h, a*ncc),  h = d.oc,  ncc = r*r - oc.oc, oc = c - o;  needs disc >= 0.
Camera rays (scan_primary) share their origin, so t + m is formed once per sphere when the tile is staged,
  tm = fl(t + m)   with t as above for the camera centre,      key = fma(g, g, tm):
one rounding moves from the end of the chain to tm.  Every case below checks both forms (any origin can be a
camera centre).
Here both are evaluated in numpy with float32 fma emulated through float64 (products of two float32 are
exact in float64), on random and adversarial ray/sphere pairs."""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def dot(u, v):
    return fma(u[2], v[2], fma(u[1], v[1], (u[0] * v[0]).astype(f32)))


def terms(o, d, c, rr):
    oc = [(c[k] - o[k]).astype(f32) for k in range(3)]
    a = dot(d, d)
    h = dot(d, oc)
    ncc = fma(-oc[2], oc[2], fma(-oc[1], oc[1], fma(-oc[0], oc[0], rr)))
    disc = fma(h, h, (a * ncc).astype(f32))
    return oc, a, h, ncc, disc


def filt(o, d, c, rr, a, T, rsq_ulp=0):
    cp = [(c[k].astype(np.float64) - np.float64(T[k])).astype(f32) for k in range(3)]       # upload, in double
    cc = sum(cp[k].astype(np.float64) ** 2 for k in range(3))
    m = np.nextafter((rr.astype(np.float64) * (1.0 + 2.0 ** -17) - cc * (1.0 - 2.0 ** -18)).astype(f32), f32(np.inf))
    op = [(o[k] - f32(T[k])).astype(f32) for k in range(3)]                                 # kernel, per ray
    rsq = (1.0 / np.sqrt(a.astype(np.float64))).astype(f32)                                 # v_rsq_f32: within 1 ulp of this
    if rsq_ulp:
        rsq = np.nextafter(rsq, f32(np.inf) if rsq_ulp > 0 else f32(0.0))
    inv = (rsq * f32(1.0 + 2.0 ** -19)).astype(f32)
    uf = [(d[k] * inv).astype(f32) for k in range(3)]
    e = -dot(uf, op)
    w = [(f32(2.0) * op[k]).astype(f32) for k in range(3)]
    k0 = -((dot(op, op) * f32(1.0 - 2.0 ** -18)).astype(f32))
    t = fma(w[2], cp[2], fma(w[1], cp[1], fma(w[0], cp[0], k0)))
    g = fma(uf[2], cp[2], fma(uf[1], cp[1], fma(uf[0], cp[0], e)))
    tm = (t + m).astype(f32)                       # staged once per sphere for the camera centre (geo0.w)
    return (fma(g, g, t) + m).astype(f32), fma(g, g, tm)


def check(o, d, c, r, label, T=(0.0, 0.0, 0.0)):
    o = [x.astype(f32) for x in o]
    d = [x.astype(f32) for x in d]
    c = [x.astype(f32) for x in c]
    rr = (r.astype(f32) * r.astype(f32)).astype(f32)
    oc, a, h, ncc, disc = terms(o, d, c, rr)
    accepted = disc >= 0
    assert accepted.sum() > 0, label
    for rsq_ulp in (-1, 0, 1):
        q, q5 = filt(o, d, c, rr, a, [f32(x) for x in T], rsq_ulp)
        for form, key in (("scan_tile", q), ("scan_primary", q5)):
            dropped = accepted & np.signbit(key)
            assert not dropped.any(), (f"{label}: {dropped.sum()} exact-test candidates dropped by the {form} filter "
                                       f"(1/sqrt off by {rsq_ulp} ulp)")
    return accepted.mean()


def unit(rng, n):
    v = rng.normal(size=(3, n))
    return v / np.linalg.norm(v, axis=0)


def test_random_pairs():
    rng = np.random.default_rng(1)
    n = 2_000_000
    c = rng.uniform(-20, 20, (3, n))
    r = rng.uniform(0.05, 2.0, n)
    o = rng.uniform(-25, 25, (3, n))
    # aim near the sphere so a useful share of the pairs is close to tangent
    aim = c + unit(rng, n) * r * rng.uniform(0.0, 1.5, n)
    d = (aim - o) * rng.uniform(0.1, 3.0, n)
    check(list(o), list(d), list(c), r, "random")
    check(list(o), list(d), list(c), r, "random, translated", T=(3.25, -7.5, 11.0))


def test_grazing_rays_exactly_at_the_rim():
    rng = np.random.default_rng(2)
    n = 2_000_000
    c = rng.uniform(-15, 15, (3, n))
    r = rng.uniform(0.1, 1.0, n)
    o = c + unit(rng, n) * rng.uniform(1.5, 60.0, n)
    oc = c - o
    dist = np.linalg.norm(oc, axis=0)
    w = oc / dist
    t = np.cross(w.T, unit(rng, n).T).T
    t /= np.linalg.norm(t, axis=0)
    # tangent direction: angle asin(r/dist) off the centre line, +- a few ulps
    sin_a = (r / dist) * (1.0 + rng.uniform(-4e-7, 4e-7, n))
    d = w * np.sqrt(np.maximum(0.0, 1.0 - sin_a ** 2)) + t * sin_a
    d *= rng.uniform(0.2, 5.0, n)
    share = check(list(o), list(d), list(c), r, "grazing")
    assert 0.2 < share < 0.8


def test_huge_ground_sphere_and_origins_on_surfaces():
    rng = np.random.default_rng(3)
    n = 2_000_000
    # the r=1000 ground sphere of the book scene seen from points on and just above it
    c = np.zeros((3, n)); c[1] = -1000.0
    r = np.full(n, 1000.0)
    o = np.stack([rng.uniform(-15, 15, n), rng.uniform(-1e-4, 3.0, n) * (rng.random(n) < 0.5), rng.uniform(-15, 15, n)])
    d = unit(rng, n)
    d[1] = np.where(rng.random(n) < 0.5, rng.uniform(-1e-3, 1e-3, n), d[1])
    d *= rng.uniform(0.5, 2.0, n)
    check(list(o), list(d), list(c), r, "ground")
    check(list(o), list(d), list(c), r, "ground, book-scene translation", T=(0.3, 0.2, -0.4))
    # origins on the surface of small spheres (bounce rays), all directions
    c = rng.uniform(-12, 12, (3, n)); r = rng.uniform(0.2, 1.0, n)
    o = c + unit(rng, n) * r * (1.0 + rng.uniform(-2e-7, 2e-7, n))
    check(list(o), list(unit(rng, n) * rng.uniform(0.3, 2.0, n)), list(c), r, "surface origins")


def test_extreme_scales():
    rng = np.random.default_rng(4)
    n = 1_000_000
    for scale, dscale in ((1e-3, 1e-3), (1e4, 1e3), (1.0, 1e-6), (1.0, 1e6), (3e5, 1.0)):
        # (the translation only matters for speed: far from T the slack makes more spheres candidates)
        c = rng.uniform(-20, 20, (3, n)) * scale
        r = rng.uniform(0.05, 2.0, n) * scale
        o = rng.uniform(-25, 25, (3, n)) * scale
        aim = c + unit(rng, n) * r * rng.uniform(0.9, 1.1, n)
        d = (aim - o) / np.linalg.norm(aim - o, axis=0) * dscale
        check(list(o), list(d), list(c), r, f"scale {scale} dir {dscale}")
        check(list(o + 1000.0 * scale), list(d), list(c + 1000.0 * scale), r, f"scale {scale} dir {dscale}, far from T")


def test_camera_inside_spheres_on_surfaces_and_far_from_the_translation():
    """The cases where the one moved rounding of the camera-ray form could matter: the camera centre inside a
    sphere (every ray hits), exactly on a surface, and a scene whose spheres lie 1e3 .. 1e5 from the translation T
    (large |c'|: large cancelling terms in t + m)."""
    rng = np.random.default_rng(5)
    n = 1_000_000
    c = rng.uniform(-10, 10, (3, n)); r = rng.uniform(0.5, 30.0, n)
    o = c + unit(rng, n) * r * rng.uniform(0.0, 0.999, n)                   # inside
    assert check(list(o), list(unit(rng, n) * rng.uniform(0.1, 4.0, n)), list(c), r, "camera inside") == 1.0
    o = c + unit(rng, n) * r                                                  # on the surface, to rounding
    check(list(o), list(unit(rng, n)), list(c), r, "camera on a surface")
    for far in (1e3, 1e4, 1e5):
        c = rng.uniform(-20, 20, (3, n)) + far * unit(rng, 1)
        r = rng.uniform(0.05, 2.0, n)
        o = c + unit(rng, n) * r * rng.uniform(1.0, 40.0, n)
        aim = c + unit(rng, n) * r * rng.uniform(0.9, 1.1, n)
        d = (aim - o) * rng.uniform(0.2, 3.0, n)
        check(list(o), list(d), list(c), r, f"centres {far:g} from T")
        # and with T where the kernel would put it (the median of the centres): the same pairs, small |c'|
        check(list(o), list(d), list(c), r, f"centres {far:g} from the origin, T at their median", T=tuple(np.median(c, axis=1)))
