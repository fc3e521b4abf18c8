"""The scan's conservative filter must never drop a sphere the exact test can accept.

Kernel (rt_kernels.hip, scan_tile), with T the scene translation, c' = fl(c - T), o' = fl(o - T):
  t = fma(w.z,c'z, fma(w.y,c'y, fma(w.x,c'x, k0)))     w = 2 o',  k0 = -(o'.o' (1-2^-18))
  g = fma(uf.z,c'z, fma(uf.y,c'y, fma(uf.x,c'x, e)))   uf = d rsq(d.d) (1+2^-19),  e = -uf.o'
                                                       rsq = the hardware's 1/sqrt, accurate to 1 ulp: modelled here as the
                                                       correctly rounded value and its two neighbours (all three are checked)
  key = fma(g, g, t) + m                               m = r*r(1+2^-17) - |c'|^2 (1-2^-18), rounded up;
  candidate iff sign(key) clear.
Exact test (checker order):  disc = fma(h, h, a*ncc),  h = d.oc,  ncc = r*r - oc.oc, oc = c - o;  needs disc >= 0.
Bounce rays of the camera-ray-batch kernels (scan_tile<true>) use  key = fma(gc, gc, t) + m  with gc = clamp(g, 0, 1):
a sphere whose centre lies behind the ray is dropped unless the origin is inside it.  There the exact test's full
precondition is the bar: disc >= 0 and (h > 0 or ncc > 0) (accept_exact).  The whole expression is evaluated times
sigma^2 (uf, e times sigma; w, k0, m times sigma^2), sigma a power of two with sigma |c - o| <= 1/2 for origins on the
scene's spheres, so that the clamp's upper end is never reached; both the scaling (bit-identical signs) and the bound
are checked here.
Camera rays (scan_primary) share their origin o and their directions are affine in the image coordinates,
d = P0 + fx du + fy dv, so with q = c - o the condition becomes |d.q| / |d| >= sqrt(|q|^2 - r*r) and d.q = C + fx A + fy B:
  rt_camera_terms (double precision, once per launch and sphere): A = du.q, B = dv.q, C = P0.q,
      sq = sqrt(|q|^2 (1-2^-18) - r*r (1+2^-17)) (1-2^-20) - s_abs, rounded down   (s_abs: see the kernel's comment)
  scan:  dq = clamp(fma(fx, A, fma(fy, B, C)), 0, 1);  key = fma(dq, rsq(d.d)(1+2^-19), -sq);  candidate iff sign(key) clear,
  with A, B, C, sq all times rho = 2^-k, rho * 2 max M < 1 (M = |C| + (W+1)|A| + (H+1)|B| bounds |dq|): the clamp's lower end
  drops spheres behind the camera ray unless the camera is inside them; the bar is accept_exact's full precondition.
The direction the exact test sees is make_ray's rounded one; the filter works from fx, fy.
Pixel units (unit_list): the spheres a unit's camera-ray batches scan are chosen once per unit by a geometric test --
centre within r_eff + |q| s + 2^-20 (|o'| + |c'|) of the line through the centre of the unit's footprint (|q x u|, single
precision), and not behind it by more than that (r_eff^2 = r*r (1+2^-17) + 2^-18 |q|^2; s = reach / |d_centre| + 2^-18,
reach from the launcher, make_launch).  Every sphere accept_exact can accept for any ray of the unit must be listed.
Block boxes (rt_upload_scene, scan_tile): a bounce ray's scan skips a block of 32 spheres unless its half-line reaches the
block's box, grown for the ray; every sphere accept_exact can accept must lie in a block whose box the ray reaches.
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


def filt(o, d, c, rr, a, T, rsq_ulp=0, sigma=1.0, half=False):
    """key of scan_tile<half> with the scene scale sigma (a power of two); also returns sigma * g, the clamp's input."""
    sigma = f32(sigma)
    s2 = (sigma * sigma).astype(f32)
    cp = [(c[k].astype(np.float64) - np.float64(T[k])).astype(f32) for k in range(3)]       # upload, in double
    cc = sum(cp[k].astype(np.float64) ** 2 for k in range(3))
    m = np.nextafter((rr.astype(np.float64) * (1.0 + 2.0 ** -17) - cc * (1.0 - 2.0 ** -18)).astype(f32), f32(np.inf))
    m = (m * s2).astype(f32)
    op = [(o[k] - f32(T[k])).astype(f32) for k in range(3)]                                 # kernel, per ray
    rsq = (1.0 / np.sqrt(a.astype(np.float64))).astype(f32)                                 # v_rsq_f32: within 1 ulp of this
    if rsq_ulp:
        rsq = np.nextafter(rsq, f32(np.inf) if rsq_ulp > 0 else f32(0.0))
    inv = ((rsq * f32(1.0 + 2.0 ** -19)).astype(f32) * sigma).astype(f32)
    uf = [(d[k] * inv).astype(f32) for k in range(3)]
    e = -dot(uf, op)
    w = [((f32(2.0) * s2).astype(f32) * op[k]).astype(f32) for k in range(3)]
    k0 = (-((dot(op, op) * f32(1.0 - 2.0 ** -18)).astype(f32)) * s2).astype(f32)
    t = fma(w[2], cp[2], fma(w[1], cp[1], fma(w[0], cp[0], k0)))
    g = fma(uf[2], cp[2], fma(uf[1], cp[1], fma(uf[0], cp[0], e)))
    gc = np.clip(g, f32(0.0), f32(1.0)) if half else g
    return (fma(gc, gc, t) + m).astype(f32), g


def scene_sigma(c, r, T):
    """The launcher's choice (rt_upload_scene): sigma = 2^-k, 2^k > 4 R, R = max |c - T| + r."""
    R = (np.sqrt(sum((c[k].astype(np.float64) - T[k]) ** 2 for k in range(3))) + r).max()
    return float(2.0 ** -np.frexp(4.0 * R)[1]), R


def check(o, d, c, r, label, T=(0.0, 0.0, 0.0), bounce=False):
    """bounce: the origins lie on (or inside) spheres of the scene {c, r}, as bounce rays' do: the half-line filter of
    scan_tile<true> with the scene's scale is checked as well."""
    o = [x.astype(f32) for x in o]
    d = [x.astype(f32) for x in d]
    c = [x.astype(f32) for x in c]
    rr = (r.astype(f32) * r.astype(f32)).astype(f32)
    oc, a, h, ncc, disc = terms(o, d, c, rr)
    accepted = disc >= 0
    assert accepted.sum() > 0, label
    sigma, R = scene_sigma(c, r, T)
    for rsq_ulp in (-1, 0, 1):
        key, _ = filt(o, d, c, rr, a, [f32(x) for x in T], rsq_ulp)
        dropped = accepted & np.signbit(key)
        assert not dropped.any(), (f"{label}: {dropped.sum()} exact-test candidates dropped by the scan_tile filter "
                                   f"(1/sqrt off by {rsq_ulp} ulp)")
        # the scene's power-of-two scale changes no sign
        scaled, _ = filt(o, d, c, rr, a, [f32(x) for x in T], rsq_ulp, sigma=sigma)
        assert np.array_equal(np.signbit(scaled), np.signbit(key)), label
        if bounce:
            needed = accepted & ((h > 0) | (ncc > 0))            # accept_exact's precondition
            assert needed.sum() > 0, label
            half, g = filt(o, d, c, rr, a, [f32(x) for x in T], rsq_ulp, sigma=sigma, half=True)
            assert g.max() <= 0.5 + 1e-3, (label, float(g.max()))  # far from the clamp's upper end
            dropped = needed & np.signbit(half)
            assert not dropped.any(), (f"{label}: {dropped.sum()} spheres the exact test can accept dropped by the half-line filter "
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
    check(list(o), list(unit(rng, n) * rng.uniform(0.3, 2.0, n)), list(c), r, "surface origins", bounce=True)
    check(list(o), list(unit(rng, n) * rng.uniform(0.3, 2.0, n)), list(c), r, "surface origins, translated", T=(0.5, -0.25, 1.0), bounce=True)


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


def test_origins_inside_spheres_on_surfaces_and_far_from_the_translation():
    """Origins inside a sphere (every ray hits), exactly on a surface, and scenes whose spheres lie 1e3 .. 1e5 from the
    translation T (large |c'|: large cancelling terms in the expansion)."""
    rng = np.random.default_rng(5)
    n = 1_000_000
    c = rng.uniform(-10, 10, (3, n)); r = rng.uniform(0.5, 30.0, n)
    o = c + unit(rng, n) * r * rng.uniform(0.0, 0.999, n)                   # inside
    assert check(list(o), list(unit(rng, n) * rng.uniform(0.1, 4.0, n)), list(c), r, "origin inside", bounce=True) == 1.0
    o = c + unit(rng, n) * r                                                  # on the surface, to rounding
    check(list(o), list(unit(rng, n)), list(c), r, "origin on a surface", bounce=True)
    for far in (1e3, 1e4, 1e5):
        c = rng.uniform(-20, 20, (3, n)) + far * unit(rng, 1)
        r = rng.uniform(0.05, 2.0, n)
        o = c + unit(rng, n) * r * rng.uniform(1.0, 40.0, n)
        aim = c + unit(rng, n) * r * rng.uniform(0.9, 1.1, n)
        d = (aim - o) * rng.uniform(0.2, 3.0, n)
        check(list(o), list(d), list(c), r, f"centres {far:g} from T")
        # and with T where the kernel would put it (the median of the centres): the same pairs, small |c'|
        check(list(o), list(d), list(c), r, f"centres {far:g} from the origin, T at their median", T=tuple(np.median(c, axis=1)))


# ---------------------------------------------------------------- camera rays (scan_primary, rt_camera_terms)
def _cam_arrays(cam):
    b = cam.block
    g = lambda v: np.array(list(v), dtype=f32)
    return g(b.center), g(b.pixel00_loc), g(b.pixel_delta_u), g(b.pixel_delta_v), int(b.image_width), int(b.image_height)


def camera_ray(cam, fx, fy):
    """make_ray of the kernel: pc = pixel00 + fma(du, fx, dv * fy); d = pc - centre (single precision)."""
    o, p00, du, dv, _, _ = _cam_arrays(cam)
    d = []
    for k in range(3):
        off = fma(np.full_like(fx, du[k]), fx, (np.full_like(fy, dv[k]) * fy).astype(f32))
        pc = (p00[k] + off).astype(f32)
        d.append((pc - o[k]).astype(f32))
    return [np.full_like(fx, o[k]) for k in range(3)], d


def camera_terms(cam, c, rr):
    """rt_camera_terms, operation for operation, in double precision; results rounded as the kernel rounds them."""
    o, p00, du, dv, W, H = [x.astype(np.float64) if isinstance(x, np.ndarray) else x for x in _cam_arrays(cam)]
    q = [c[k].astype(np.float64) - o[k] for k in range(3)]
    p0 = p00 - o
    A = sum(du[k] * q[k] for k in range(3))
    B = sum(dv[k] * q[k] for k in range(3))
    C = sum(p0[k] * q[k] for k in range(3))
    qq = sum(q[k] * q[k] for k in range(3))
    W1, H1 = W + 1.0, H + 1.0
    nrm = np.cross(du, dv)
    F = abs(np.dot(p0, nrm)) / np.linalg.norm(nrm)
    eta = np.sqrt(sum((2.0 ** -24 * (abs(p00[k]) + abs(p0[k]) + 3.0 * W1 * abs(du[k]) + 4.0 * H1 * abs(dv[k]))) ** 2 for k in range(3)))
    M = np.abs(C) + W1 * np.abs(A) + H1 * np.abs(B)
    s_abs = (4.0 * 2.0 ** -24 * M + 2.0 * np.sqrt(qq) * eta) / F + 2.0 ** -22 * np.sqrt(qq)
    ntm = qq * (1.0 - 2.0 ** -18) - rr.astype(np.float64) * (1.0 + 2.0 ** -17)
    with np.errstate(invalid="ignore"):
        v = np.sqrt(np.maximum(ntm, 0.0)) * (1.0 - 2.0 ** -20) - s_abs
    v = np.where((ntm > 0.0) & (v > 0.0), v, -np.inf)
    # no bound left, but the camera is not inside by the exact test's own ncc (single precision, sphere_terms' order):
    # the half-space test alone, C shifted by delta, the smallest positive sq
    o32 = _cam_arrays(cam)[0]
    oc32 = [(c[k] - o32[k]).astype(f32) for k in range(3)]
    ncc = fma(-oc32[2], oc32[2], fma(-oc32[1], oc32[1], fma(-oc32[0], oc32[0], rr)))
    d_max = np.linalg.norm(p0) + W1 * np.linalg.norm(du) + H1 * np.linalg.norm(dv)
    delta = 2.0 * (4.0 * 2.0 ** -24 * M + np.sqrt(qq) * eta + 3.0 * 2.0 ** -24 * d_max * np.sqrt(qq))
    half = np.isneginf(v) & (ncc <= 0)
    C = np.where(half, C + delta, C)
    M = np.where(half, M + delta, M)
    rho = 2.0 ** -np.frexp(2.0 * M.max())[1] if np.isfinite(M.max()) and M.max() > 0 else 1.0   # the whole scene's scale
    v = v * rho
    sq = v.astype(f32)
    sq = np.where(np.isfinite(v) & (sq.astype(np.float64) > v), np.nextafter(sq, f32(-np.inf)), sq).astype(f32)
    sq = np.where(half, f32(2.0 ** -126), sq).astype(f32)
    return (A * rho).astype(f32), (B * rho).astype(f32), (C * rho).astype(f32), sq


def check_camera(cam, fx, fy, c, r, label):
    fx, fy = fx.astype(f32), fy.astype(f32)
    c = [x.astype(f32) for x in c]
    rr = (r.astype(f32) * r.astype(f32)).astype(f32)
    o, d = camera_ray(cam, fx, fy)
    _, a, h, ncc, disc = terms(o, d, c, rr)
    A, B, C, sq = camera_terms(cam, c, rr)
    accepted = (disc >= 0) & ((h > 0) | (ncc > 0))               # accept_exact's precondition
    assert accepted.sum() > 0, label
    dq = fma(fx, A, fma(fy, B, C))
    assert np.abs(dq).max() <= 0.5 + 1e-3, (label, float(np.abs(dq).max()))   # far from the clamp's upper end
    for rsq_ulp in (-1, 0, 1):
        rsq = (1.0 / np.sqrt(a.astype(np.float64))).astype(f32)
        if rsq_ulp:
            rsq = np.nextafter(rsq, f32(np.inf) if rsq_ulp > 0 else f32(0.0))
        ils = (rsq * f32(1.0 + 2.0 ** -19)).astype(f32)
        with np.errstate(invalid="ignore"):
            key = fma(np.clip(dq, f32(0.0), f32(1.0)), ils, -sq)
        dropped = accepted & np.signbit(key)
        assert not dropped.any(), f"{label}: {dropped.sum()} exact-test candidates dropped by the camera-ray filter (1/sqrt off by {rsq_ulp} ulp)"
    return accepted, ~np.signbit(key)


def _cameras():
    import ray_tracing_amd as rt
    return {
        "book 4K": rt.Camera.look_at(16 / 9, 3840, 20.0, (13, 2, 3), (0, 0, 0)),
        "hall 8K": rt.Camera.look_at(16 / 9, 7680, 90.0, (0, 0, -85), (0, 0, -70)),
        "grid": rt.Camera.look_at(16 / 9, 1920, 40.0, (0, 30, 80), (0, 0, 0)),
        "cpu crate 1920": rt.Camera.new_cpu(16 / 9, 1920),
        "cpu crate 256": rt.Camera.new_cpu(16 / 9, 256),
        "far from the origin": rt.Camera.look_at(16 / 9, 1280, 35.0, (10013, 2002, -4997), (10000, 2000, -5000)),
        "very far from the origin": rt.Camera.look_at(4 / 3, 640, 60.0, (3.0e5, -1.0e5, 2.0e5), (3.0e5 + 3, -1.0e5, 2.0e5 + 4)),
        "narrow": rt.Camera.look_at(1.0, 512, 0.5, (0, 0, 50), (0, 0, 0)),
    }


def test_camera_ray_filter_grazing_and_random():
    """Rays through random image points (pixel + jitter) against spheres placed so that the ray grazes them within a
    few ulps, passes well inside, or misses; cameras of every benchmark configuration and some hostile ones."""
    rng = np.random.default_rng(6)
    n = 400_000
    for name, cam in _cameras().items():
        o_, p00, du, dv, W, H = _cam_arrays(cam)
        fx = rng.integers(0, W, n) + rng.uniform(-0.5, 1.0, n)
        fy = rng.integers(0, H, n) + rng.uniform(-0.5, 1.0, n)
        _, d = camera_ray(cam, fx.astype(f32), fy.astype(f32))
        d64 = np.stack([x.astype(np.float64) for x in d])
        dl = np.linalg.norm(d64, axis=0)
        u = d64 / dl
        perp = np.cross(u.T, unit(rng, n).T).T
        perp /= np.linalg.norm(perp, axis=0)
        scale = max(1.0, float(np.linalg.norm(p00.astype(np.float64) - o_.astype(np.float64))))
        r = rng.uniform(0.05, 2.0, n) * rng.choice([1.0, 1.0, 30.0], n)
        dist = rng.uniform(0.2, 80.0, n) * scale * rng.choice([1.0, -1.0], n, p=[0.9, 0.1])   # some centres behind the camera
        for kind, miss in (("grazing", 1.0 + rng.uniform(-4e-7, 4e-7, n)), ("random", rng.uniform(0.0, 1.5, n))):
            c = o_.astype(np.float64)[:, None] + u * dist + perp * r * miss
            acc, cand = check_camera(cam, fx, fy, list(c), r, f"{name}, {kind}")
            assert acc.mean() > 0.05, (name, kind, acc.mean())
            if kind == "random" and "far" not in name:
                # the filter stays tight where scenes live (spheres within a few focal lengths, at least half a degree across): few listed in vain
                near = (np.abs(dist) < 3.0 * scale) & (r > 0.01 * np.abs(dist))
                assert cand[near].mean() < acc[near].mean() + 0.05, (name, cand[near].mean(), acc[near].mean())


def test_camera_ray_filter_camera_inside_on_surfaces_and_the_ground_sphere():
    rng = np.random.default_rng(7)
    n = 300_000
    for name, cam in _cameras().items():
        o_, p00, du, dv, W, H = _cam_arrays(cam)
        o64 = o_.astype(np.float64)[:, None]
        fx = rng.integers(0, W, n) + rng.uniform(-0.5, 1.0, n)
        fy = rng.integers(0, H, n) + rng.uniform(-0.5, 1.0, n)
        r = rng.uniform(0.5, 30.0, n)
        c = o64 + unit(rng, n) * r * rng.uniform(0.0, 0.999, n)                         # camera inside: every ray hits
        assert check_camera(cam, fx, fy, list(c), r, f"{name}, camera inside")[0].mean() > (0.5 if "far" in name else 0.999)
        c = o64 + unit(rng, n) * r                                                        # camera on the surface, to rounding
        acc, cand = check_camera(cam, fx, fy, list(c), r, f"{name}, camera on a surface")
        if "far" not in name:                                                             # half of these spheres lie behind their ray: not listed
            assert cand.mean() < 0.8, (name, cand.mean())
        # a huge sphere just below the camera (the book scene's ground), rays near its horizon included
        r = np.full(n, 1000.0)
        c = o64 + np.stack([rng.uniform(-5, 5, n), -1000.0 - rng.uniform(1e-4, 3.0, n), rng.uniform(-5, 5, n)])
        check_camera(cam, fx, fy, list(c), r, f"{name}, ground sphere")


def unit_reach(cam, uw, uh):
    """make_launch (rt_hip.cpp): (w + 1/2)/2 |du| + (h + 1/2)/2 |dv| + eta, rounded up to single precision."""
    o, p00, du, dv, W, H = [x.astype(np.float64) if isinstance(x, np.ndarray) else x for x in _cam_arrays(cam)]
    W1, H1 = W + 1.0, H + 1.0
    eta = np.sqrt(sum((2.0 ** -24 * (abs(p00[k]) + abs(p00[k] - o[k]) + 3.0 * W1 * abs(du[k]) + 4.0 * H1 * abs(dv[k]))) ** 2 for k in range(3)))
    reach = (0.5 * (uw + 0.5) * np.linalg.norm(du) + 0.5 * (uh + 0.5) * np.linalg.norm(dv) + eta) * (1.0 + 2.0 ** -20)
    return np.nextafter(f32(reach), f32(np.inf))


def unit_bundle(cam, x0, y0, uw, uh, T=(0.0, 0.0, 0.0)):
    """The per-unit part of unit_list: central direction u, o' = o - T, the bundle's sine s and 2^-20 |o'| (1-norm)."""
    o, p00, du, dv, W, H = [x.astype(np.float64) if isinstance(x, np.ndarray) else x for x in _cam_arrays(cam)]
    T = np.array(T, dtype=f32)
    reach = unit_reach(cam, uw, uh)
    fxc = x0.astype(np.float64) + 0.5 * (uw - 0.5)
    fyc = y0.astype(np.float64) + 0.5 * (uh - 0.5)
    d = [(p00[k] - o[k]) + (fxc * du[k] + fyc * dv[k]) for k in range(3)]
    len2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
    down = lambda x: np.nextafter(np.nextafter(x, f32(0.0)), f32(0.0))
    up = lambda x: np.nextafter(np.nextafter(x, f32(np.inf)), f32(np.inf))
    len_lo = (up(np.sqrt(len2.astype(f32))) * f32(1.0 - 2.0 ** -20)).astype(f32)
    assert (len_lo > reach).all()
    s_ = ((reach / len_lo).astype(f32) * f32(1.0 + 2.0 ** -20)).astype(f32) + f32(2.0 ** -18)
    inv = 1.0 / np.sqrt(len2)
    u = [(d[k] * inv).astype(f32) for k in range(3)]
    op = [np.full_like(u[0], f32(o[k] - np.float64(T[k]))) for k in range(3)]
    o_abs = (f32(2.0 ** -20) * ((np.abs(op[0]) + np.abs(op[1])).astype(f32) + np.abs(op[2])).astype(f32)).astype(f32)
    return u, op, s_, o_abs


def unit_listed(cam, x0, y0, uw, uh, c, rr, T=(0.0, 0.0, 0.0)):
    """unit_list of the kernel for the unit at (x0, y0) (arrays: one unit and one sphere per element), operation for
    operation: the unit's central direction u, o' = o - T and s in double precision, rounded to single; per sphere in
    single precision q = c' - o' (c' = c - T as uploaded), the distance to the central line as |q x u|, and
    bound = r_eff + |q| s + 2^-20 (|o'| + |c'|).  The hardware's 1-ulp square roots are modelled 2 ulps too SMALL where
    a larger value lists more, too large where it lists less (the model is then stricter than the kernel)."""
    u, op, s_, o_abs = unit_bundle(cam, x0, y0, uw, uh, T)
    T = np.array(T, dtype=f32)
    down = lambda x: np.nextafter(np.nextafter(x, f32(0.0)), f32(0.0))
    cp = [(c[k].astype(np.float64) - np.float64(T[k])).astype(f32) for k in range(3)]           # upload, in double
    q = [(cp[k] - op[k]).astype(f32) for k in range(3)]
    qq = (fma(q[2], q[2], fma(q[1], q[1], (q[0] * q[0]).astype(f32))) * f32(1.0 + 2.0 ** -20)).astype(f32)
    t = fma(u[2], q[2], fma(u[1], q[1], (u[0] * q[0]).astype(f32)))
    kx = fma(q[1], u[2], -(q[2] * u[1]).astype(f32))
    ky = fma(q[2], u[0], -(q[0] * u[2]).astype(f32))
    kz = fma(q[0], u[1], -(q[1] * u[0]).astype(f32))
    dist2 = fma(kz, kz, fma(ky, ky, (kx * kx).astype(f32)))
    val = (fma(rr, np.full_like(rr, f32(1.0 + 2.0 ** -17)), (f32(2.0 ** -18) * qq).astype(f32)) * f32(1.0 + 2.0 ** -20)).astype(f32)
    r_eff = (down(np.sqrt(val)) * f32(1.0 + 2.0 ** -20)).astype(f32)
    q_len = (down(np.sqrt(qq)) * f32(1.0 + 2.0 ** -20)).astype(f32)
    c_abs = ((np.abs(cp[0]) + np.abs(cp[1])).astype(f32) + np.abs(cp[2])).astype(f32)
    bound = ((fma(q_len, s_.astype(f32), r_eff) + fma(np.full_like(c_abs, f32(2.0 ** -20)), c_abs, o_abs)).astype(f32) * f32(1.0 + 2.0 ** -20)).astype(f32)
    return ~(dist2 > (bound * bound).astype(f32)) & ~(t < -bound)


def test_unit_lists_hold_every_sphere_a_ray_of_the_unit_can_hit():
    """Rays anywhere in a unit's footprint (corners and edges included) against spheres that graze them within a few
    ulps, are hit squarely, are missed, contain the camera or lie behind it: what accept_exact can accept is on the
    unit's list -- for the unit shapes the launcher uses, on every benchmark camera and the hostile ones -- and the
    lists stay short: spheres that a ray misses by more than the unit's width and their own radius are not listed."""
    rng = np.random.default_rng(9)
    n = 300_000
    for name, cam in _cameras().items():
        o_, p00, du, dv, W, H = _cam_arrays(cam)
        for uw, uh in ((16, 1), (4, 1), (16, 4), (8, 8)):
            x0 = (rng.integers(0, max(1, W // uw), n) * uw).astype(np.int64)
            y0 = (rng.integers(0, max(1, H // uh), n) * uh).astype(np.int64)
            # image coordinates as sample_ray makes them: pixel + [-1/2, 1/2) or the grid sampler's [0, 1); a third on the footprint's rim
            fx = x0 + rng.integers(0, uw, n) + rng.uniform(-0.5, 1.0, n)
            fy = y0 + rng.integers(0, uh, n) + rng.uniform(-0.5, 1.0, n)
            rim = rng.random(n) < 0.33
            fx = np.where(rim, np.where(rng.random(n) < 0.5, x0 - 0.5, x0 + uw - 2.0 ** -10), fx)
            fy = np.where(rim, np.where(rng.random(n) < 0.5, y0 - 0.5, y0 + uh - 2.0 ** -10), fy)
            fx, fy = fx.astype(f32), fy.astype(f32)
            o, d = camera_ray(cam, fx, fy)
            d64 = np.stack([x.astype(np.float64) for x in d])
            dl = np.linalg.norm(d64, axis=0)
            u = d64 / dl
            perp = np.cross(u.T, unit(rng, n).T).T
            perp /= np.linalg.norm(perp, axis=0)
            scale = max(1.0, float(np.linalg.norm(p00.astype(np.float64) - o_.astype(np.float64))))
            r = rng.uniform(0.05, 2.0, n) * rng.choice([1.0, 1.0, 30.0], n)
            dist = rng.uniform(0.2, 80.0, n) * scale * rng.choice([1.0, -1.0], n, p=[0.9, 0.1])
            for kind, miss in (("grazing", 1.0 + rng.uniform(-4e-7, 4e-7, n)), ("random", rng.uniform(0.0, 1.5, n)),
                               ("camera inside", None)):
                if miss is None:
                    r2 = rng.uniform(0.5, 30.0, n)
                    c = o_.astype(np.float64)[:, None] + unit(rng, n) * r2 * rng.uniform(0.0, 1.0, n)
                    rr = (r2.astype(f32) * r2.astype(f32)).astype(f32)
                else:
                    c = o_.astype(np.float64)[:, None] + u * dist + perp * r * miss
                    rr = (r.astype(f32) * r.astype(f32)).astype(f32)
                c32 = [x.astype(f32) for x in c]
                _, a, h, ncc, disc = terms(o, d, c32, rr)
                accepted = (disc >= 0) & ((h > 0) | (ncc > 0))
                listed = unit_listed(cam, x0, y0, uw, uh, c32, rr)
                label = f"{name}, {uw}x{uh} units, {kind}"
                assert accepted.sum() > 0, label
                dropped = accepted & ~listed
                assert not dropped.any(), f"{label}: {dropped.sum()} spheres a ray of the unit can hit are not on the unit's list"
                # the scene's translation T only moves roundings: where the launcher would put it (among the spheres) and far off
                for T in (tuple(np.median(c, axis=1)), tuple(o_.astype(np.float64) + 1000.0 * scale)):
                    moved = unit_listed(cam, x0, y0, uw, uh, c32, rr, T=T)
                    assert not (accepted & ~moved).any(), f"{label}, T = {T}: spheres a ray of the unit can hit are not on the unit's list"
                if kind == "random" and "far" not in name:
                    # a sphere missed by three radii and thrice the footprint's width at its depth is not listed
                    foot = np.abs(dist) * float(unit_reach(cam, uw, uh)) / dl
                    clear = (miss > 1.4) & (dist > 0) & (0.4 * r > 3.0 * foot) & (r > 2.0 ** -7 * np.abs(dist))   # (r_eff adds 2^-9 |q| in quadrature)
                    if clear.sum() > 100:
                        assert listed[clear].mean() < 0.01, (label, listed[clear].mean())


def block_box(cp, r, R):
    """rt_upload_scene: the box of a block of spheres (arrays [members, blocks]) in coordinates relative to T: every
    sphere's own |r| (1 + 2^-10) + 2^-9 |c'| + 2^-19 R beyond its centre, rounded outwards."""
    cc = sum(cp[k].astype(np.float64) ** 2 for k in range(3))
    ext = np.abs(r.astype(np.float64)) * (1.0 + 2.0 ** -10) + 2.0 ** -9 * np.sqrt(cc) + 2.0 ** -19 * R
    lo = [np.nextafter((cp[k].astype(np.float64) - ext).astype(f32), f32(-np.inf)).min(axis=0) for k in range(3)]
    hi = [np.nextafter((cp[k].astype(np.float64) + ext).astype(f32), f32(np.inf)).max(axis=0) for k in range(3)]
    return lo, hi

def block_ball(lo, hi):
    """rt_upload_scene: the ball round a block's box -- centre rounded to single precision, radius half the diagonal
    plus the centre's rounding, times 1 + 2^-7, rounded up."""
    mid = [0.5 * (lo[k].astype(np.float64) + hi[k].astype(np.float64)) for k in range(3)]
    d2 = sum((hi[k].astype(np.float64) - lo[k].astype(np.float64)) ** 2 for k in range(3))
    c1 = sum(np.abs(m) for m in mid)
    rb = (0.5 * np.sqrt(d2) * (1.0 + 2.0 ** -40) + 2.0 ** -22 * c1) * (1.0 + 2.0 ** -7)
    return [m.astype(f32) for m in mid], np.nextafter(rb.astype(f32), f32(np.inf))


def ball_reached(cam, x0, y0, uw, uh, cb, rb, T=(0.0, 0.0, 0.0)):
    """unit_list's test of a block's ball, operation for operation (square roots 2 ulps too small: stricter than the kernel)."""
    u, op, s_, o_abs = unit_bundle(cam, x0, y0, uw, uh, T)
    down = lambda x: np.nextafter(np.nextafter(x, f32(0.0)), f32(0.0))
    q = [(cb[k] - op[k]).astype(f32) for k in range(3)]
    q_len = (down(np.sqrt(fma(q[2], q[2], fma(q[1], q[1], (q[0] * q[0]).astype(f32))))) * f32(1.0 + 2.0 ** -18)).astype(f32)
    t = fma(u[2], q[2], fma(u[1], q[1], (u[0] * q[0]).astype(f32)))
    kx = fma(q[1], u[2], -(q[2] * u[1]).astype(f32))
    ky = fma(q[2], u[0], -(q[0] * u[2]).astype(f32))
    kz = fma(q[0], u[1], -(q[1] * u[0]).astype(f32))
    dist2 = fma(kz, kz, fma(ky, ky, (kx * kx).astype(f32)))
    c_abs = ((np.abs(cb[0]) + np.abs(cb[1])).astype(f32) + np.abs(cb[2])).astype(f32)
    first = fma((q_len + rb).astype(f32), (s_.astype(f32) + f32(2.0 ** -8)).astype(f32), rb)
    bound = ((first + fma(np.full_like(c_abs, f32(2.0 ** -19)), c_abs, (f32(2.0) * o_abs).astype(f32))).astype(f32) * f32(1.0 + 2.0 ** -18)).astype(f32)
    return ~(dist2 > (bound * bound).astype(f32)) & ~(t < -bound)


def test_unit_lists_of_streamed_scenes_reach_the_block_of_every_listed_sphere():
    """A streamed scene's unit lists look at a block's spheres only if the ball round the block's box passes: for every
    sphere the sphere test lists (grazing, squarely hit, camera inside, behind), in blocks of neighbours from a tenth to
    ten times its size around it, the block's ball passes too; and balls far off the unit's line do not."""
    rng = np.random.default_rng(10)
    n, members = 100_000, 6
    for name, cam in _cameras().items():
        o_, p00, du, dv, W, H = _cam_arrays(cam)
        for uw, uh in ((8, 2), (4, 1), (16, 1)):
            x0 = (rng.integers(0, max(1, W // uw), n) * uw).astype(np.int64)
            y0 = (rng.integers(0, max(1, H // uh), n) * uh).astype(np.int64)
            fx = (x0 + rng.uniform(-0.5, uw, n)).astype(f32)
            fy = (y0 + rng.uniform(-0.5, uh, n)).astype(f32)
            o, d = camera_ray(cam, fx, fy)
            d64 = np.stack([x.astype(np.float64) for x in d])
            u = d64 / np.linalg.norm(d64, axis=0)
            perp = np.cross(u.T, unit(rng, n).T).T
            perp /= np.linalg.norm(perp, axis=0)
            scale = max(1.0, float(np.linalg.norm(p00.astype(np.float64) - o_.astype(np.float64))))
            r = rng.uniform(0.05, 2.0, n) * rng.choice([1.0, 1.0, 30.0], n)
            dist = rng.uniform(0.2, 80.0, n) * scale * rng.choice([1.0, -1.0], n, p=[0.9, 0.1])
            miss = np.where(rng.random(n) < 0.5, 1.0 + rng.uniform(-4e-7, 4e-7, n), rng.uniform(0.0, 6.0, n))
            c0 = o_.astype(np.float64)[:, None] + u * dist + perp * r * miss
            inside = rng.random(n) < 0.1                                   # the camera inside the sphere
            c0 = np.where(inside, o_.astype(np.float64)[:, None] + unit(rng, n) * r * rng.uniform(0.0, 1.0, n), c0)
            spread = r * np.exp(rng.uniform(np.log(0.1), np.log(10.0), n))
            for T in ((0.0, 0.0, 0.0), tuple(np.median(c0, axis=1)), tuple(o_.astype(np.float64) + 1000.0 * scale)):
                # the block: the sphere and its neighbours, [members, n] per axis, as uploaded (relative to T)
                cs = [np.stack([c0[k]] + [c0[k] + spread * rng.uniform(-1.0, 1.0, n) for _ in range(members - 1)]) for k in range(3)]
                rs = np.stack([r] + [r * rng.uniform(0.2, 1.5, n) for _ in range(members - 1)]).astype(f32)
                c32 = [x.astype(f32) for x in cs]
                cp = [(c32[k].astype(np.float64) - np.float64(f32(T[k]))).astype(f32) for k in range(3)]
                R = float(max(np.abs(np.stack(cp)).max() * 2.0, 1.0))
                lo, hi = block_box(cp, rs, R)
                cb, rb = block_ball(lo, hi)
                reached = ball_reached(cam, x0, y0, uw, uh, cb, rb, T=T)
                label = f"{name}, {uw}x{uh} units, T = {T}"
                some = 0
                for m in range(members):
                    listed = unit_listed(cam, x0, y0, uw, uh, [c32[k][m] for k in range(3)], (rs[m] * rs[m]).astype(f32), T=T)
                    some += int(listed.sum())
                    lost = listed & ~reached
                    assert not lost.any(), f"{label}: {lost.sum()} listed spheres in blocks whose ball does not pass"
                assert some > 1000, label
                if T == (0.0, 0.0, 0.0) and "far" not in name:
                    far_off = (miss > 4.5) & ~inside     # (the ball: sqrt 3 x 1.7 r)
                    far_off = far_off & (dist > 0) & (spread < 0.2 * r) & (r > 2.0 ** -5 * np.abs(dist)) & \
                              (r > 8.0 * np.abs(dist) * float(unit_reach(cam, uw, uh)) / np.linalg.norm(d64, axis=0))
                    if far_off.sum() > 100:
                        assert reached[far_off].mean() < 0.05, (label, reached[far_off].mean())


# ---------------------------------------------------------------- block boxes (rt_upload_scene, scan_tile)

def box_reached(lo, hi, op, d, rcp_ulp=0, sigma=1.0):
    """scan_tile's slab test, operation for operation: the box grown by 2^-9 |o'| (1 + 2^-8) for this ray; a plane's
    parameter fma(plane, inv, -(o' +- grow) inv) with inv = 1 / uf clamped to +-2^40 / sigma, uf = d rsq(d.d) (1 + 2^-19)
    sigma the filter's scaled unit direction; minNum / maxNum, the half-line t >= 0; reached unless tf < tn.  The
    hardware's reciprocal and 1/sqrt are within 1 ulp (rcp_ulp moves both), its square root is modelled 2 ulps too small
    (a smaller box); a reciprocal's denormal argument counts as zero."""
    down = lambda x: np.nextafter(np.nextafter(x, f32(0.0)), f32(0.0))
    sigma = f32(sigma)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
        a = dot(d, d)
        rsq = (1.0 / np.sqrt(a.astype(np.float64))).astype(f32)
        if rcp_ulp:
            rsq = np.nextafter(rsq, f32(np.inf) if rcp_ulp > 0 else f32(0.0))
        inv_len = ((rsq * f32(1.0 + 2.0 ** -19)).astype(f32) * sigma).astype(f32)
        uf = [(d[k] * inv_len).astype(f32) for k in range(3)]
        uf = [np.where(np.abs(x) < f32(2.0 ** -126), np.copysign(f32(0.0), x), x).astype(f32) for x in uf]
        inv = [(f32(1.0) / uf[k]).astype(f32) for k in range(3)]
        if rcp_ulp:
            inv = [np.nextafter(x, np.where(x > 0, f32(np.inf), f32(-np.inf)).astype(f32) if rcp_ulp > 0 else f32(0.0)) for x in inv]
        lim = (f32(2.0 ** 40) * (f32(1.0) / sigma)).astype(f32)
        inv = [np.where(np.isnan(x), -lim, np.clip(x, -lim, lim)).astype(f32) for x in inv]
        grow = (down(np.sqrt(dot(op, op))) * f32(2.0 ** -9 * (1.0 + 2.0 ** -8))).astype(f32)
        n_lo = [(-((op[k] + grow).astype(f32)) * inv[k]).astype(f32) for k in range(3)]
        n_hi = [(-((op[k] - grow).astype(f32)) * inv[k]).astype(f32) for k in range(3)]
        a = [fma(np.broadcast_to(lo[k], inv[k].shape).astype(f32), inv[k], n_lo[k]) for k in range(3)]
        b = [fma(np.broadcast_to(hi[k], inv[k].shape).astype(f32), inv[k], n_hi[k]) for k in range(3)]
        tn = np.fmax(np.fmax(np.fmin(a[0], b[0]), np.fmin(a[1], b[1])), np.fmax(np.fmin(a[2], b[2]), f32(0.0)))
        tf = np.fmin(np.fmin(np.fmax(a[0], b[0]), np.fmax(a[1], b[1])), np.fmax(a[2], b[2]))
        return ~(tf < tn)


def test_block_boxes_are_reached_by_every_ray_that_can_hit_a_sphere_of_the_block():
    """Blocks of 32 spheres (clusters as the launcher's sort makes them, a flat layer resting on a ground sphere, one
    huge member among small ones, scenes far from the translation T), bounce rays from origins on and inside spheres of
    the scene, aimed to graze a member within a few ulps, to hit it squarely or to miss -- among them directions with
    components that are exactly zero, denormal or 1e-30 (a degenerate diffuse bounce: the case a product-minus-product
    slab parameter gets wrong) and origins exactly on a face of the box.  Whatever accept_exact can accept lies in a
    block whose box the ray's half-line reaches; and the boxes are worth having: most rays that miss a block's members
    by a few block diameters do not reach its box."""
    rng = np.random.default_rng(10)
    n, m = 120_000, 32
    for label, extent, layer, far in (("clusters", 12.0, False, 0.0), ("flat layer", 12.0, True, 0.0), ("far from T", 12.0, False, 3000.0),
                                      ("tiny scene", 0.02, False, 0.0)):
        centre = rng.uniform(-extent, extent, (3, 1, n))
        size = extent * rng.uniform(0.02, 0.3, (1, 1, n))
        c = centre + rng.uniform(-1.0, 1.0, (3, m, n)) * size
        r = size[0] * np.exp(rng.uniform(np.log(0.02), np.log(0.5), (m, n)))
        if layer:
            c[1] = r                                                           # resting on the plane y = 0
            c[1, 0] = -1000.0 * extent; r[0] = 1000.0 * extent                 # ... which is a huge sphere, member 0 of every block
            c[0, 0] = 0.0; c[2, 0] = 0.0
        c = c + far
        T = np.array([far, far, far]) + rng.uniform(-0.3, 0.3, 3) * extent
        cp = [(c[k] - T[k]).astype(f32) for k in range(3)]                     # upload: c - T in double, rounded
        c32 = [c[k].astype(f32) for k in range(3)]
        R = float((np.sqrt(sum(cp[k].astype(np.float64) ** 2 for k in range(3))) + r).max())
        lo, hi = block_box(cp, r.astype(f32), R)
        # one target member per ray; origin on (or in) another sphere of the scene, a few block sizes away or right next to it
        j = rng.integers(0, m, n)
        idx = (j, np.arange(n))
        tc = np.stack([c32[k][idx].astype(np.float64) for k in range(3)])
        tr = r.astype(f32)[idx].astype(np.float64)
        dist = size[0, 0] * np.exp(rng.uniform(np.log(0.05), np.log(60.0), n))
        o = tc + unit(rng, n) * (tr + dist)
        inside = rng.random(n) < 0.1
        o = np.where(inside, tc + unit(rng, n) * tr * rng.uniform(0.0, 1.0, n), o)
        o32 = [o[k].astype(f32) for k in range(3)]
        for kind, miss in (("grazing", 1.0 + rng.uniform(-4e-7, 4e-7, n)), ("random", rng.uniform(0.0, 3.0, n))):
            oc = tc - np.stack([x.astype(np.float64) for x in o32])
            dl = np.linalg.norm(oc, axis=0)
            w = oc / np.maximum(dl, 1e-300)
            t = np.cross(w.T, unit(rng, n).T).T
            t /= np.maximum(np.linalg.norm(t, axis=0), 1e-300)
            sin_a = np.minimum((tr / np.maximum(dl, tr)) * miss, 1.0)
            d = (w * np.sqrt(np.maximum(0.0, 1.0 - sin_a ** 2)) + t * sin_a) * rng.uniform(0.2, 3.0, n)
            d32 = [d[k].astype(f32) for k in range(3)]
            # degenerate directions: one component exactly 0, -0, denormal or 1e-30; the origin's matching coordinate
            # sometimes exactly on a face of the box
            k0 = rng.integers(0, 3, n)
            special = rng.random(n) < 0.3
            val = rng.choice(np.array([0.0, -0.0, 1e-41, -1e-41, 1e-30, -1e-30], dtype=f32), n)
            for k in range(3):
                sel = special & (k0 == k)
                d32[k] = np.where(sel, val, d32[k]).astype(f32)
                on_face = sel & (rng.random(n) < 0.2)
                o32[k] = np.where(on_face, (lo[k] + f32(T[k])).astype(f32), o32[k]).astype(f32)
            _, a, h, ncc, disc = terms(o32, d32, [c32[k][idx] for k in range(3)], (r.astype(f32)[idx] * r.astype(f32)[idx]).astype(f32))
            accepted = (disc >= 0) & ((h > 0) | (ncc > 0))
            assert accepted.sum() > 1000, (label, kind)
            op = [(o32[k] - f32(T[k])).astype(f32) for k in range(3)]
            for rcp_ulp in (-1, 0, 1):
                reached = box_reached(lo, hi, op, d32, rcp_ulp, sigma=2.0 ** -np.frexp(4.0 * R)[1])
                dropped = accepted & ~reached
                assert not dropped.any(), f"{label}, {kind}: {dropped.sum()} rays that can hit a sphere of a block do not reach the block's box"
        if not layer:
            # ... and worth having: rays from outside aimed four block sizes past the block do not reach its box
            away = (dist > 8.0 * size[0, 0]) & ~inside
            aim = centre[:, 0, :] + unit(rng, n) * 4.0 * 1.75 * size[0, 0]
            d_off = [(aim[k] - o[k]).astype(f32) for k in range(3)]
            op = [(o[k].astype(f32) - f32(T[k])).astype(f32) for k in range(3)]
            assert box_reached(lo, hi, op, d_off, sigma=2.0 ** -np.frexp(4.0 * R)[1])[away].mean() < 0.05, label


def test_half_line_filter_on_bounce_rays_between_spheres():
    """Bounce rays as the kernels see them: origins on the surface of one sphere of a scene (to rounding, inside and
    outside), aimed at, past and away from another sphere of the same scene -- grazing it within a few ulps, hitting it
    squarely, or with the target behind the ray; the book scene's ground and the mirror hall's r = 25 mirrors among
    them.  The half-line filter must keep every sphere accept_exact can accept, and drops most of those behind."""
    rng = np.random.default_rng(8)
    n = 1_000_000
    for name, extent, big in (("book", 12.0, (0.0, -1000.0, 0.0, 1000.0)), ("hall", 6.0, (0.0, 0.0, 30.0, 25.0)), ("plain", 20.0, None)):
        # a scene of m spheres; pairs (origin sphere i, target sphere j) drawn from it
        m = 400
        sc = rng.uniform(-extent, extent, (3, m)); sr = rng.uniform(0.2, 1.0, m)
        if big:
            sc[:, 0] = big[:3]; sr[0] = big[3]
        T = np.median(sc, axis=1)
        i = rng.integers(0, m, n); j = rng.integers(0, m, n)
        i[: n // 4] = 0 if big else i[: n // 4]                    # a quarter of the rays leave the big sphere
        j[n // 4: n // 2] = 0 if big else j[n // 4: n // 2]        # a quarter aim at it
        on = unit(rng, n)
        if big:                                                    # on the big sphere: near the part the scene sits on
            toward = (np.median(sc[:, 1:], axis=1) - sc[:, 0])
            toward /= np.linalg.norm(toward)
            on[:, i == 0] = (toward[:, None] + 0.02 * unit(rng, int((i == 0).sum())))
            on /= np.linalg.norm(on, axis=0)
        o = sc[:, i] + on * sr[i] * (1.0 + rng.uniform(-3e-7, 3e-7, n))
        c, r = sc[:, j], sr[j]
        oc = c - o
        dist = np.maximum(np.linalg.norm(oc, axis=0), 1e-9)
        wdir = oc / dist
        perp = np.cross(wdir.T, unit(rng, n).T).T
        perp /= np.maximum(np.linalg.norm(perp, axis=0), 1e-30)
        kind = rng.integers(0, 4, n)
        sin_a = np.clip(r / dist, 0.0, 1.0) * np.where(kind == 0, 1.0 + rng.uniform(-4e-7, 4e-7, n), rng.uniform(0.0, 1.3, n))
        sin_a = np.clip(sin_a, 0.0, 1.0)
        d = wdir * np.sqrt(1.0 - sin_a ** 2) + perp * sin_a
        d = np.where(kind == 3, -d, d)                             # target behind the ray
        d = np.where(kind == 2, unit(rng, n), d)                   # any direction
        d *= rng.uniform(0.3, 3.0, n)
        # the filter is evaluated with the whole scene's scale: pass the scene for sigma, the pairs for the test
        o32 = [x.astype(f32) for x in o]; d32 = [x.astype(f32) for x in d]; c32 = [x.astype(f32) for x in c]
        rr = (r.astype(f32) * r.astype(f32)).astype(f32)
        _, a, h, ncc, disc = terms(o32, d32, c32, rr)
        needed = (disc >= 0) & ((h > 0) | (ncc > 0))
        sigma, R = scene_sigma([x.astype(f32) for x in sc], sr, T)
        for rsq_ulp in (-1, 0, 1):
            half, g = filt(o32, d32, c32, rr, a, [f32(x) for x in T], rsq_ulp, sigma=sigma, half=True)
            assert g.max() <= 0.5 + 1e-3, (name, float(g.max()))
            assert not (needed & np.signbit(half)).any(), (name, rsq_ulp, int((needed & np.signbit(half)).sum()))
            full, _ = filt(o32, d32, c32, rr, a, [f32(x) for x in T], rsq_ulp, sigma=sigma)
        listed_full, listed_half = (~np.signbit(full)).mean(), (~np.signbit(half)).mean()
        assert needed.mean() > 0.05 and listed_half < listed_full, (name, needed.mean(), listed_half, listed_full)
        # behind the ray and origin outside: the full-line filter lists them, the half-line filter almost none
        behind = (kind == 3) & (i != j) & (ncc < 0)
        assert (~np.signbit(full))[behind].mean() > 0.3 and (~np.signbit(half))[behind].mean() < 0.02, name
