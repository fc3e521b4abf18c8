"""The C-ABI library loads without a GPU and exports every symbol include/rt_hip.h declares; host-side helpers
(camera set-up, scene builders, PPM writer) against the CPU checker and the reference's constants.
No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rt():
    import ray_tracing_amd
    return ray_tracing_amd


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(rt):
    names = declared_functions()
    assert len(names) >= 20
    lib = C.CDLL(rt._ffi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rt_hip.h but not exported"
    # and the binding table covers exactly the header
    assert sorted(rt._ffi.SYMBOLS) == names


def test_struct_layouts_match_reference_abi(rt):
    """Offsets of the 148-byte camera block (ray_tracing.rs:251-272 / library.cu:54-75) and the 36-byte sphere."""
    cam = rt.RtCamera
    want = {"aspect_ratio": 0, "image_width": 4, "samples_per_pixel": 8, "max_depth": 12, "vfov": 16, "lookfrom": 20,
            "lookat": 32, "vup": 44, "image_height": 56, "pixel_samples_scale": 60, "center": 64, "pixel00_loc": 76,
            "pixel_delta_u": 88, "pixel_delta_v": 100, "u": 112, "v": 124, "w": 136}
    for name, off in want.items():
        assert getattr(cam, name).offset == off, name
    assert C.sizeof(cam) == 148
    assert rt.SPHERE_DTYPE.itemsize == 36
    assert C.sizeof(rt.RtParams) == 40


def test_create_without_gpu_returns_an_error_not_a_crash(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError) as e:
        rt.HipWorld(0)
    assert "hip" in str(e.value).lower()


def test_look_at_camera_equals_checker(rt, oracle):
    """rt_camera_initialize vs the checker's restatement of ray_tracing.rs:300-336: same bytes."""
    for (w, vfov, frm, at) in [(1280, 90.0, (-2, 2, 1), (0, 0, -1)), (3840, 20.0, (13, 2, 3), (0, 0, 0)),
                               (257, 35.5, (0.5, -3, 9), (1, 1, 1)), (64, 90.0, (0, 0, -85), (0, 0, -70))]:
        cam = rt.Camera.look_at(16.0 / 9.0, w, vfov, frm, at)
        ref = oracle.RtCamera()
        C.memmove(C.addressof(ref), C.addressof(cam.block), 148)
        for f in ("pixel00_loc", "pixel_delta_u", "pixel_delta_v", "u", "v", "w", "center"):
            getattr(ref, f)[:] = [0.0, 0.0, 0.0]
        ref.image_height = 0
        oracle.lib().or32_camera_initialize(C.addressof(ref))
        assert bytes(ref) == bytes(cam.block)
    d = rt.Camera.new(16.0 / 9.0, 1280)           # reference defaults, ray_tracing.rs:276-294
    assert d.image_height == 720 and d.block.vfov == 90.0 and list(d.block.lookfrom) == [-2.0, 2.0, 1.0]
    assert d.block.samples_per_pixel == 10 and d.block.max_depth == 10


def test_cpu_camera_equals_checker(rt, oracle):
    """rt_camera_new_cpu vs the checker's f64 Camera::new (lib.rs:148-183) rounded to f32."""
    for w in (256, 1920, 5, 1000):
        cam = rt.Camera.new_cpu(16.0 / 9.0, w)
        ref = oracle.camera_f32_from_d(oracle.camera_new_f64(16.0 / 9.0, w))
        got = oracle.camera_f32_from_rt(cam.block)
        assert bytes(ref) == bytes(got)
    assert rt.Camera.new_cpu(16.0 / 9.0, 256).image_height == 144
    with pytest.raises(rt.RtError):
        rt.Camera.new_cpu(16.0 / 9.0, 0)


def test_scene_builders(rt):
    cpu = rt.scene_cpu_default()                    # main.rs:26-28
    assert len(cpu) == 2 and list(cpu[0]["center"]) == [0.0, 0.0, -1.0] and cpu[1]["radius"] == 100.0
    gpu = rt.scene_gpu_default()                    # ray.cu:97-106
    assert len(gpu) == 9
    assert list(gpu["material"]) == [1, 1, 2, 2, 2, 2, 2, 1, 2]
    assert gpu[3]["param"] == 1.0 and np.isclose(gpu[2]["param"], 0.3) and gpu[5]["radius"] == 25.0
    b = rt.scene_book1_final(1)
    assert 470 <= len(b) <= 488 and b[0]["radius"] == 1000.0
    assert np.array_equal(b, rt.scene_book1_final(1)) and not np.array_equal(b, rt.scene_book1_final(2))
    small = b[1:-3]
    assert (small["radius"] == np.float32(0.2)).all() and set(np.unique(small["material"])) == {1, 2, 3}
    assert (np.hypot(small["center"][:, 0] - 4.0, small["center"][:, 2]) > 0.9).all()
    assert (b[b["material"] == 2]["param"] <= 1.0).all() and (b[b["material"] == 3]["param"] == 1.5).all()
    g = rt.scene_grid(2, 128)
    assert len(g) == 16385 and (g[1:]["radius"] < 0.45).all()
    assert len(rt.scene_grid(2, 4)) == 17
    h = rt.scene_mirror_hall(3, 64)
    assert len(h) == 66 and h[0]["radius"] == 25.0 and h[1]["center"][2] == -110.0
    # Reflect(color, fuzz) clamps fuzz to 1 (library.cu:144)
    assert rt.Reflect((1, 1, 1), 7.0)[2] == 1.0
    s = rt.Sphere((1, 2, 3), 0.5, rt.Dielectric(1.5))
    assert s["material"] == 3 and s["param"] == 1.5
    w = rt.world([rt.Sphere((0, 0, -1), 0.5), s])
    assert w.dtype == rt.SPHERE_DTYPE and len(w) == 2 and w[0]["material"] == 1


def test_ppm_writer(rt, tmp_path):
    img = np.zeros((3, 5, 4), np.uint8)
    img[..., 0] = np.arange(5)[None, :] * 50
    img[..., 1] = np.arange(3)[:, None] * 100
    img[..., 2] = 7
    img[..., 3] = 255
    path = tmp_path / "x.ppm"
    rt.write_ppm(path, img)
    data = open(path, "rb").read()
    assert data.startswith(b"P6\n5 3\n255\n")
    body = np.frombuffer(data[len(b"P6\n5 3\n255\n"):], np.uint8).reshape(3, 5, 3)
    assert np.array_equal(body, img[..., :3])
    with pytest.raises(rt.RtError):
        rt.write_ppm("/nonexistent-dir/x.ppm", img)


def test_p3_float_writer_has_the_reference_image_format(rt, tmp_path):
    """cpu-implementation/image.ppm: `P3`, `256 144`, `255`, then `r g b` lines of unrounded 255.999*colour with
    shortest round-trip digits (its sky pixels end in `255.999`)."""
    lin = np.zeros((2, 3, 4), np.float32)
    lin[..., 2] = 1.0
    lin[0, 0, :2] = (0.5, 0.25)
    lin[1, 2, :2] = (np.float32(0.1), 1.0)
    path = tmp_path / "x.ppm"
    rt.write_ppm_p3f(path, lin)
    lines = open(path).read().split("\n")
    assert lines[:3] == ["P3", "3 2", "255"] and len(lines) == 3 + 6 + 1 and lines[-1] == ""
    assert lines[3] == "127.9995 63.99975 255.999"
    vals = [[float(t) for t in l.split()] for l in lines[3:9]]
    want = 255.999 * lin[..., :3].astype(np.float64).reshape(-1, 3)
    assert np.array_equal(np.array(vals), want)          # exact: shortest digits round-trip
    assert lines[8].split()[0] == repr(255.999 * float(np.float32(0.1)))


def test_p3_float_writer_never_prints_an_exponent(rt, tmp_path):
    """Rust's `{}` of an f64 (what wrote cpu-implementation/image.ppm) has no scientific notation: a dark pixel
    such as 255.999 * 1e-7 is written 0.0000255999..., not 2.55999e-05; whole numbers have no fraction; a NaN
    is `NaN`."""
    lin = np.zeros((1, 4, 4), np.float32)
    lin[0, 0, :3] = (1e-7, 1e-12, 0.0)
    lin[0, 1, :3] = (3.0e-39, 1.0, 1e9)            # a subnormal float, and a large value
    lin[0, 2, :3] = (np.nan, np.inf, -np.inf)
    lin[0, 3, :3] = (-1e-7, 0.5, 0.25)
    path = tmp_path / "dark.ppm"
    rt.write_ppm_p3f(path, lin)
    lines = open(path).read().split("\n")
    body = lines[3:7]
    assert not any("e" in l.lower().replace("inf", "").replace("nan", "") for l in body), body
    first = body[0].split()
    assert first[0].startswith("0.0000255999") and first[1].startswith("0.0000000002559") and first[2] == "0"
    assert [float(t) for t in first] == [255.999 * float(np.float32(1e-7)), 255.999 * float(np.float32(1e-12)), 0.0]
    second = [float(t) for t in body[1].split()]
    assert second == [255.999 * float(np.float32(3.0e-39)), 255.999, 255.999 * float(np.float32(1e9))]
    assert body[2].split() == ["NaN", "inf", "-inf"]
    assert body[3].split()[0].startswith("-0.0000255999")


def test_missing_rccl_is_an_error_code_not_a_crash(rt, tmp_path):
    """rt_comm_id / rt_comm_init with an RCCL that cannot be loaded return RT_ERR_COMM and leave a message; the
    process survives.  Runs in a child process because the library binds RCCL once per process."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "import ray_tracing_amd._ffi as f\n"
        "L = f.lib()\n"
        "buf = (C.c_uint8 * 128)()\n"
        "rc = L.rt_comm_id(C.byref(buf))\n"
        "msg = (L.rt_last_error(None) or b'').decode()\n"
        "print(rc, '|', msg)\n"
        "rc2 = L.rt_comm_id(C.byref(buf))\n"
        "sys.exit(0 if (rc == 10005 and rc2 == 10005 and 'cannot load RCCL' in msg and 'no-such-rccl' in msg) else 1)\n"
    ) % ROOT
    env = dict(os.environ, RT_RCCL_LIB=str(tmp_path / "no-such-rccl.so"))
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)


def test_command_line_renderer_is_built():
    exe = os.path.join(ROOT, "ray_tracing_amd", "csrc", "rt_render")
    assert os.path.exists(exe) and os.access(exe, os.X_OK)


def test_rust_crate_declares_only_exported_symbols():
    """rust/ray_tracing_hip/src/ffi.rs (source only: no Rust toolchain here) must not drift from the C header:
    every function it declares is exported by the library, and its three structs have the header's sizes."""
    import re
    import ray_tracing_amd._ffi as ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "rust", "ray_tracing_hip", "src", "ffi.rs")).read()
    names = re.findall(r"pub fn (rt_[a-z0-9_]+)\(", text)
    assert len(names) >= 20
    for n in names:
        assert n in ffi.SYMBOLS, n
    def words(struct):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, text, re.S).group(1)
        total = 0
        for ty in re.findall(r"pub [a-z0-9_]+: ([^,]+),", body):
            m = re.match(r"\[(f32|u32); (\d+)\]", ty)
            total += {"f32": 4, "u32": 4, "u64": 8}[m.group(1) if m else ty] * (int(m.group(2)) if m else 1)
        return total
    assert words("RtSphere") == 36 and words("RtCamera") == 148 and words("RtParams") == 40


def test_fake_rccl_exports_what_the_library_binds():
    """tests/native/libfake_rccl.so (test infrastructure for the multi-rank GPU test) must offer every RCCL entry
    point rt_hip.cpp resolves at run time, or the library would refuse to load it."""
    import ctypes as C
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "ray_tracing_amd", "csrc", "rt_hip.cpp")).read()
    bound = set(re.findall(r'sym\("(nccl[A-Za-z]+)"\)', src))
    assert {"ncclSend", "ncclRecv", "ncclAllReduce", "ncclCommAbort"} <= bound
    path = os.path.join(root, "tests", "native", "libfake_rccl.so")
    assert os.path.exists(path), "make -C tests/native"
    nm = subprocess_output(["nm", "-D", "--defined-only", path])
    for name in bound:
        assert re.search(r"\bT %s\b" % name, nm), name


def subprocess_output(cmd):
    import subprocess
    return subprocess.run(cmd, capture_output=True, text=True, check=True).stdout


def test_bench_starts_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher: too few GPUs is a one-line error and exit code 2; with enough
    GPUs it starts torch's launcher as a child process with N ranks on 127.0.0.1 and passes its own flags on."""
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import torch
    args = types.SimpleNamespace(gpus=4)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench.spawn_ranks(args) == 2
    assert "needs 4 GPUs, found 1" in capsys.readouterr().err
    seen = {}
    import subprocess
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.spawn_ranks(args) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
