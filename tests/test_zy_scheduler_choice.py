"""The library's choice between its two schedulers for sample_chunk = 1 (pixel units held per wavefront:
rt_render_queue; pooled per workgroup: rt_render_pool) is a rule on the launch size (rt_hip.cpp, make_launch: short launches -> pooled).  This
walks launch sizes across the rule's threshold -- the whole frame, then rank 0's interleaved row tiles of 2, 4, 8 and 16
ranks -- on the two frames the rule was fitted on (book scene 3840x2160 at 500 spp, depth 50; mirror hall 7680x4320 at
256 spp, depth 100), times the render kernel under each forced scheduler and under the library's own choice (HIP events
inside the library, best of two launches), and asserts that the scheduler it chooses is within 2 % (+ 0.2 ms) of the faster one, and
that the three frames are the same bit for bit.  Runs late (file name) because it is the longest GPU test (~20 s)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_library_picks_the_faster_scheduler_across_launch_sizes():
    torch = pytest.importorskip("torch")
    import ray_tracing_amd as rt
    from ray_tracing_amd import tiles
    cases = [("book", rt.scene_book1_final(1), rt.Camera.look_at(16.0 / 9.0, 3840, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0)), 500, 50),
             ("hall", rt.scene_mirror_hall(3, 64), rt.Camera.look_at(16.0 / 9.0, 7680, 90.0, (0.0, 0.0, -85.0), (0.0, 0.0, -70.0)), 256, 100)]
    report, late = [], []
    for name, sp, cam, spp, depth in cases:
        H, W = cam.image_height, cam.image_width
        p = rt.DeviceGUI(samples_per_pixel=spp, max_depth=depth, seed=1, sample_chunk=1)
        with rt.HipWorld(0, sp) as hip:
            buf = tiles.local_buffer(H, W, 1, "cuda")
            torch.cuda.synchronize()
            for world in (1, 2, 4, 8, 16):
                n = len(tiles.tiles_of_rank(H, 0, world))
                ms, kernel, sums, rays = {}, {}, {}, {}
                for sched in ("wave", "pool", "auto"):
                    hip.debug_schedule(scheduler=sched)
                    best = float("inf")
                    for _ in range(2):
                        r, t = hip.render_tiles_device(cam, p, tiles.TILE_ROWS, 0, world, n, buf.data_ptr())
                        best = min(best, t)
                    ms[sched], kernel[sched], rays[sched] = best, hip.last_kernel_name(), r
                    sums[sched] = int(buf[: n * tiles.TILE_ROWS].to(torch.int64).sum().item())
                hip.debug_schedule()
                assert kernel["wave"] == "rt_render_queue_boxes"      # (boxes: scenes of 32 spheres and more)
                assert kernel["pool"] == "rt_render_pool_boxes"
                assert rays["wave"] == rays["pool"] == rays["auto"] and sums["wave"] == sums["pool"] == sums["auto"]
                fastest = min(ms["wave"], ms["pool"])
                report.append(f"{name} 1/{world}: wave {ms['wave']:.1f} ms, pool {ms['pool']:.1f} ms, library ({kernel['auto']}) {ms['auto']:.1f} ms")
                # the library's CHOICE is judged by that scheduler's own forced timing (a third timing of the same kernel
                # differs from the first by up to 4 % on the small launches)
                chosen = "pool" if kernel["auto"].startswith("rt_render_pool") else "wave"
                if ms[chosen] > 1.02 * fastest + 0.2:
                    late.append(report[-1])
    print("\n".join(report))
    assert not late, late
