"""rt_render_tiles_device_static: the entry point that allocates nothing (counter block in the library's static
device array) against rt_render_tiles_device (counter block from hipMalloc), through the C-ABI, bit for bit:
RGBA8, linear radiance and segment count, on two calls in a row, the second with a smaller input, for the
kernels that bench.py and the fold mode run.  Also: a code object that states a counter block larger than the
static array is rejected by the static entry point only."""
import os

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
