"""rt_render_tiles_device_static: the entry point that allocates nothing (counter block in the library's static
device array) against rt_render_tiles_device (counter block from hipMalloc), through the C-ABI, bit for bit:
RGBA8, linear radiance and segment count, on two calls in a row, the second with a smaller input, for the
kernels that bench.py and the fold mode run.  Also: a code object that states a counter block larger than the
static array is rejected by the static entry point only.  And the claim itself is observed: free device memory is
the same before and after a static call, and in a child process whose HIP allocation calls are counted
(tests/native/hip_alloc_count.c, loaded ahead of the runtime) the count does not move across static calls while it does
move across an allocating one."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def render_both(torch, rt, hip, cam, p, tile_rows, first_tile, stride, n_tiles):
    W = cam.image_width
    out = []
    for static in (False, True):
        d_rgba = torch.full((n_tiles * tile_rows, W, 4), 99, dtype=torch.uint8, device="cuda")
        d_lin = torch.full((n_tiles * tile_rows, W, 4), -7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        rays, _ = hip.render_tiles_device(cam, p, tile_rows, first_tile, stride, n_tiles, d_rgba.data_ptr(),
                                          d_lin.data_ptr(), static_scratch=static)
        out.append((d_rgba.cpu().numpy(), d_lin.cpu().numpy(), rays))
    (rgba0, lin0, rays0), (rgba1, lin1, rays1) = out
    assert rays1 == rays0 and rays0 > 0
    assert np.array_equal(rgba1, rgba0)
    assert np.array_equal(lin1.view(np.uint32), lin0.view(np.uint32))
    return rgba1, rays1


def test_static_scratch_entry_point_matches_the_allocating_one():
    torch = pytest.importorskip("torch")
    import ray_tracing_amd as rt
    sp = rt.scene_book1_final(1)
    with rt.HipWorld(0, sp) as hip:
        # first call: 200x112 frame, all 14 row tiles; second call: a smaller frame and only every other tile
        big = rt.Camera.look_at(16.0 / 9.0, 200, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0))
        small = rt.Camera.look_at(16.0 / 9.0, 72, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0))      # H = 40
        for chunk in (1, 0, 3):          # rt_render_queue, rt_render in fold order, rt_render with runs of 3
            p = rt.DeviceGUI(samples_per_pixel=9, max_depth=20, seed=5, sample_chunk=chunk)
            rgba_big, rays_big = render_both(torch, rt, hip, big, p, 8, 0, 1, 14)
            rgba_small, rays_small = render_both(torch, rt, hip, small, p, 8, 1, 2, 2)
            assert rays_small < rays_big
            # and the static entry point's frame is the one rt_render_rows returns
            ref = hip.render(big, p)
            assert np.array_equal(rgba_big, ref["rgba"]) and rays_big == ref["rays"]
            ref = hip.render(small, p)
            assert np.array_equal(rgba_small[:8], ref["rgba"][8:16]) and np.array_equal(rgba_small[8:], ref["rgba"][24:32])
        # streamed sphere list (rt_render_queue_stream), then the small scene again
        hip.upload_scene(rt.scene_grid(2, 48))
        cam = rt.Camera.look_at(16.0 / 9.0, 96, 40.0, (0.0, 30.0, 80.0), (0.0, 0.0, 0.0))
        render_both(torch, rt, hip, cam, rt.DeviceGUI(samples_per_pixel=3, max_depth=8, seed=2, sample_chunk=1), 8, 2, 1, 3)
        hip.upload_scene(rt.scene_gpu_default())
        render_both(torch, rt, hip, small, rt.DeviceGUI(sample2_per_pixel=2, max_depth=4), 8, 0, 1, 5)
        # errors: same argument checks as the allocating entry point
        d = torch.zeros((8, 72, 4), dtype=torch.uint8, device="cuda")
        with pytest.raises(rt.RtError):
            hip.render_tiles_device(small, rt.DeviceGUI(), 12, 0, 1, 1, d.data_ptr(), static_scratch=True)
        with pytest.raises(rt.RtError):
            hip.render_tiles_device(small, rt.DeviceGUI(), 8, 5, 1, 1, d.data_ptr(), static_scratch=True)   # tile 5 starts at row 40 = H


def test_static_scratch_rejects_a_code_object_with_a_larger_counter_block():
    torch = pytest.importorskip("torch")
    import ray_tracing_amd as rt
    path = os.path.join(os.path.dirname(rt.__file__), "csrc", "rt_kernels_bigcounters.hsaco")
    cam = rt.Camera.new(16.0 / 9.0, 64)
    p = rt.DeviceGUI(sample2_per_pixel=2, max_depth=4)
    with rt.HipWorld(0, rt.scene_gpu_default()) as hip:
        d = torch.zeros((40, 64, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        want, _ = hip.render_tiles_device(cam, p, 8, 0, 1, 5, d.data_ptr(), static_scratch=True)
        hip.load_module(open(path, "rb").read())        # states 512 bytes of counters; the static array holds 256
        got, _ = hip.render_tiles_device(cam, p, 8, 0, 1, 5, d.data_ptr())
        assert got == want
        with pytest.raises(rt.RtError) as e:
            hip.render_tiles_device(cam, p, 8, 0, 1, 5, d.data_ptr(), static_scratch=True)
        assert "512" in str(e.value) and "256" in str(e.value)
        hip.load_module(None)
        got, _ = hip.render_tiles_device(cam, p, 8, 0, 1, 5, d.data_ptr(), static_scratch=True)
        assert got == want


def test_static_entry_point_leaves_free_device_memory_unchanged():
    """hipMemGetInfo (torch.cuda.mem_get_info) reads the same before and after static calls that follow a warm-up
    call -- for the exact-accumulation kernel, the fold (which stays on the kernel without a workspace here) and a
    smaller second input."""
    torch = pytest.importorskip("torch")
    import ray_tracing_amd as rt
    with rt.HipWorld(0, rt.scene_book1_final(1)) as hip:
        big = rt.Camera.look_at(16.0 / 9.0, 200, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0))
        small = rt.Camera.look_at(16.0 / 9.0, 72, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0))
        d = torch.zeros((112, 200, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for chunk in (1, 0):
            p = rt.DeviceGUI(samples_per_pixel=9, max_depth=20, seed=5, sample_chunk=chunk)
            hip.render_tiles_device(big, p, 8, 0, 1, 14, d.data_ptr(), static_scratch=True)        # warm-up
            before = torch.cuda.mem_get_info()
            hip.render_tiles_device(big, p, 8, 0, 1, 14, d.data_ptr(), static_scratch=True)
            hip.render_tiles_device(small, p, 8, 1, 2, 2, d.data_ptr(), static_scratch=True)
            assert torch.cuda.mem_get_info() == before
            assert hip.last_kernel_name() == ("rt_render_pool_plain" if chunk else "rt_render")   # (64-pixel units: no room for sphere lists)


COUNTING_WORKER = textwrap.dedent("""
    import ctypes as C, importlib.util, os, sys
    sys.path.insert(0, %(root)r)
    # the HIP runtime this process will use: PyTorch's copy when installed (what the library binds to), else ROCm's
    spec = importlib.util.find_spec("torch")
    runtime = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else ""
    if not os.path.exists(runtime):
        runtime = "/opt/rocm/lib/libamdhip64.so.7"
    os.environ["HIP_ALLOC_COUNT_RUNTIME"] = runtime
    counter = C.CDLL(%(shim)r, mode=C.RTLD_GLOBAL)            # ahead of the runtime in the global lookup order
    counter.hip_alloc_count.restype = C.c_long
    import ray_tracing_amd as rt
    hiprt = C.CDLL(runtime)                                     # direct handle: these calls are not counted
    counts = lambda: (counter.hip_alloc_count(0), counter.hip_alloc_count(1))

    def dmalloc(n):
        p = C.c_void_p()
        assert hiprt.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
        return p.value

    with rt.HipWorld(0, rt.scene_book1_final(1)) as hip:
        after_setup = counts()
        assert after_setup[0] >= 1, after_setup                # rt_upload_scene's allocation went through the counter
        big = rt.Camera.look_at(16.0 / 9.0, 200, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0))
        small = rt.Camera.look_at(16.0 / 9.0, 72, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0))
        d_rgba, d_lin = dmalloc(112 * 200 * 4), dmalloc(112 * 200 * 16)
        for chunk in (1, 0, 3):
            p = rt.DeviceGUI(samples_per_pixel=9, max_depth=20, seed=5, sample_chunk=chunk)
            before = counts()
            r1, _ = hip.render_tiles_device(big, p, 8, 0, 1, 14, d_rgba, d_lin, static_scratch=True)
            r2, _ = hip.render_tiles_device(small, p, 8, 1, 2, 2, d_rgba, d_lin, static_scratch=True)
            assert counts() == before, (chunk, before, counts())
            assert r1 > r2 > 0
        # the counter is live: the allocating entry points move it (a fold workspace, a device frame)
        before = counts()
        p = rt.DeviceGUI(samples_per_pixel=9, max_depth=20, seed=5, sample_chunk=0)
        hip.render_tiles_device(big, p, 8, 0, 1, 14, d_rgba, d_lin)
        assert hip.last_kernel_name() == "rt_render_pool_fold"
        mid = counts()
        assert mid[0] >= before[0] + 1, (before, mid)
        hip.render(big, p)
        assert counts()[0] >= mid[0] + 1
    print("no allocation observed")
""")


def test_static_entry_point_makes_no_allocation_call(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = os.path.join(root, "tests", "native", "libhip_alloc_count.so")
    assert os.path.exists(shim), "tests/native/libhip_alloc_count.so is not built (make -C tests/native)"
    script = tmp_path / "count_allocations.py"
    script.write_text(COUNTING_WORKER % {"root": root, "shim": shim})
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    assert "no allocation observed" in res.stdout
