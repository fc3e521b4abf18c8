"""The scan's conservative filter must never drop a sphere the exact test can accept.

Kernel (rt_kernels.hip, scan_tile):  q = fma(hf, hf, nf),  hf = uf.oc,  nf = rr_f - oc.oc,
  uf = d * (1/sqrt(d.d)) * (1 + 2^-19),  rr_f = r*r * (1 + 2^-17) rounded up;   candidate iff sign(q) clear.
Exact test (checker order):            disc = fma(h, h, a*ncc),  h = d.oc,  ncc = r*r - oc.oc;  needs disc >= 0.
Here both are evaluated in numpy with float32 fma emulated through float64 (products of two float32 are
exact in float64), on random and adversarial ray/sphere pairs."""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def terms(o, d, c, rr):
    oc = [(c[k] - o[k]).astype(f32) for k in range(3)]
    a = fma(d[2], d[2], fma(d[1], d[1], (d[0] * d[0]).astype(f32)))
    h = fma(d[2], oc[2], fma(d[1], oc[1], (d[0] * oc[0]).astype(f32)))
    ncc = fma(-oc[2], oc[2], fma(-oc[1], oc[1], fma(-oc[0], oc[0], rr)))
    disc = fma(h, h, (a * ncc).astype(f32))
    return oc, a, h, ncc, disc


def filt(oc, a, d, rr):
    inv = ((f32(1.0) / np.sqrt(a)).astype(f32) * f32(1.0 + 2.0 ** -19)).astype(f32)
    uf = [(d[k] * inv).astype(f32) for k in range(3)]
    rr_f = np.nextafter((rr.astype(np.float64) * (1.0 + 2.0 ** -17)).astype(f32), f32(np.inf))
    hf = fma(uf[2], oc[2], fma(uf[1], oc[1], (uf[0] * oc[0]).astype(f32)))
    nf = fma(-oc[2], oc[2], fma(-oc[1], oc[1], fma(-oc[0], oc[0], rr_f)))
    return fma(hf, hf, nf)


def check(o, d, c, r, label):
    o = [x.astype(f32) for x in o]
    d = [x.astype(f32) for x in d]
    c = [x.astype(f32) for x in c]
    rr = (r.astype(f32) * r.astype(f32)).astype(f32)
    oc, a, h, ncc, disc = terms(o, d, c, rr)
    q = filt(oc, a, d, rr)
    accepted = disc >= 0
    dropped = accepted & np.signbit(q)
    assert accepted.sum() > 0, label
    assert not dropped.any(), f"{label}: {dropped.sum()} exact-test candidates dropped by the filter"
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
    # origins on the surface of small spheres (bounce rays), all directions
    c = rng.uniform(-12, 12, (3, n)); r = rng.uniform(0.2, 1.0, n)
    o = c + unit(rng, n) * r * (1.0 + rng.uniform(-2e-7, 2e-7, n))
    check(list(o), list(unit(rng, n) * rng.uniform(0.3, 2.0, n)), list(c), r, "surface origins")


def test_extreme_scales():
    rng = np.random.default_rng(4)
    n = 1_000_000
    for scale, dscale in ((1e-3, 1e-3), (1e4, 1e3), (1.0, 1e-6), (1.0, 1e6), (3e5, 1.0)):
        c = rng.uniform(-20, 20, (3, n)) * scale
        r = rng.uniform(0.05, 2.0, n) * scale
        o = rng.uniform(-25, 25, (3, n)) * scale
        aim = c + unit(rng, n) * r * rng.uniform(0.9, 1.1, n)
        d = (aim - o) / np.linalg.norm(aim - o, axis=0) * dscale
        check(list(o), list(d), list(c), r, f"scale {scale} dir {dscale}")
