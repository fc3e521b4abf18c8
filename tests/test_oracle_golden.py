"""The CPU checker against the reference's only known-answer data (cpu-implementation/image.ppm,
cut down to tests/golden/image_ppm_fixture.npz by tests/golden/make_image_ppm_fixture.py) and against
the published PCG32 demo vectors."""
import ctypes as C
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image_ppm_fixture.npz")


@pytest.fixture(scope="module")
def golden():
    g = np.load(GOLDEN)
    W, H = int(g["width"]), int(g["height"])
    sphere = np.unpackbits(g["sphere_bits"])[:W * H].astype(bool).reshape(H, W)
    return W, H, sphere, g["index"], g["values"]


def _golden_scene():
    import oracle_ffi as o
    sp = np.zeros(1, o.SPHERE_DTYPE)
    sp[0]["center"] = (0.0, 0.0, -1.0)
    sp[0]["radius"] = 0.5
    sp[0]["material"] = 1
    return sp


def test_camera_new_matches_lib_rs_constants(oracle):
    cam = oracle.camera_new_f64(16.0 / 9.0, 256)
    assert (cam.image_width, cam.image_height) == (256, 144)
    assert list(cam.center) == [0.0, 0.0, 0.0]
    vw = 2.0 * (16.0 / 9.0)
    assert cam.pixel_delta_u[0] == vw / 256.0 and cam.pixel_delta_v[1] == -2.0 / 144.0
    assert cam.pixel00_loc[2] == -1.0
    # image_height never 0 (lib.rs:149-156)
    assert oracle.camera_new_f64(1000.0, 10).image_height == 1


def test_sky_pixels_bit_exact(oracle, golden):
    """Camera::new + pixel-centre ray + miss shading reproduce every sampled sky value exactly."""
    W, H, sphere, index, values = golden
    cam = oracle.camera_new_f64(16.0 / 9.0, W)
    L = oracle.lib()
    o3 = (C.c_double * 3)()
    d3 = (C.c_double * 3)()
    rgb = (C.c_double * 3)()
    checked = 0
    for k, v in zip(index, values):
        x, y = int(k) % W, int(k) // W
        if sphere[y, x]:
            continue
        L.or64_pixel_center_ray(C.byref(cam), x, y, o3, d3)
        L.or64_miss_color(d3, 0.5, rgb)
        got = [255.999 * rgb[0], 255.999 * rgb[1], 255.999 * rgb[2]]
        assert got == list(v), (x, y, got, list(v))
        checked += 1
    assert checked > 6000


def test_sphere_silhouette_exact_and_normals(oracle, golden):
    """Sphere::hit: the set of pixels whose centre ray hits the sphere equals the image's silhouette
    (5428 pixels, exact); hit point/normal reproduce the normal-shaded values to 1e-12 relative."""
    W, H, sphere, index, values = golden
    cam = oracle.camera_new_f64(16.0 / 9.0, W)
    idx, t, point, normal = oracle.primary_hits(cam, _golden_scene(), 0.0, float("inf"),
                                                 flags=oracle.FLAG_FLIP_SIGNBIT, details=True)
    assert (idx >= 0).sum() == 5428
    assert np.array_equal(idx >= 0, sphere)
    # the image shades with the outward normal; the record's normal is flipped to face the ray
    outward = (point - np.array([0.0, 0.0, -1.0])) / 0.5
    checked = 0
    for k, v in zip(index, values):
        x, y = int(k) % W, int(k) // W
        if not sphere[y, x]:
            continue
        got = 255.999 * 0.5 * (outward[y, x] + 1.0)
        assert np.allclose(got, v, rtol=1e-12, atol=0), (x, y, got, v)
        assert np.allclose(normal[y, x], outward[y, x])  # front face: no flip
        checked += 1
    assert checked > 1000


def test_f32_silhouette_matches_golden(oracle, golden):
    """The single-precision restatement finds the same silhouette on this scene."""
    W, H, sphere, _, _ = golden
    camf = oracle.camera_f32_from_d(oracle.camera_new_f64(16.0 / 9.0, W))
    idx = oracle.primary_hits(camf, _golden_scene(), 0.0, float("inf"))
    assert np.array_equal(idx >= 0, sphere)


def test_pcg32_published_vectors(oracle):
    """pcg32_srandom_r(42, 54) then six outputs, as printed by the PCG reference demo (pcg32-demo)."""
    L = oracle.lib()
    st, inc = C.c_uint64(0), C.c_uint64(0)
    L.or_pcg32_seed(C.byref(st), C.byref(inc), 42, 54)
    got = [L.or_pcg32_next(C.byref(st), inc.value) for _ in range(6)]
    assert got == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]


def test_u01_range(oracle):
    L = oracle.lib()
    assert L.or_u01(0) == 0.0
    assert L.or_u01(0xFFFFFFFF) == 1.0 - 2.0 ** -24
    assert L.or_u01(0x80000000) == 0.5
