"""rt_render_frame_sharded with more than one rank, on one GPU: two and three ranks run as host threads of one
process, one context each on GPU 0, and meet through tests/native (an in-process stand-in for librccl that the
library loads through RT_RCCL_LIB: ranks rendezvous on a condition variable, tiles move with hipMemcpy).  This runs
the library's own multi-rank code -- the status all-reduce, the grouped ncclSend / ncclRecv with their `part * peer`
offsets, the assembly of uneven tile sets -- which a single-GPU box can otherwise only enter as rank 0 of 1.
Asserted: rank 0's frame equals rt_render_rows bit for bit; the ranks' segment counts add up to the single launch's;
a rank that fails locally (before the exchange) or inside the exchange makes every rank return, none waits.
The child process is needed because the library binds its RCCL once per process (other GPU tests use the real one)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "native", "libfake_rccl.so")

WORKER = textwrap.dedent("""
    import ctypes as C, os, sys, threading
    sys.path.insert(0, %(root)r)
    import numpy as np
    import ray_tracing_amd as rt
    from ray_tracing_amd import _ffi

    lib = _ffi.lib()
    fake = C.CDLL(os.environ["RT_RCCL_LIB"])
    fake.fake_rccl_calls.restype = C.c_long
    RT_ERR_INVALID, RT_ERR_COMM = 10001, 10005

    def run_ranks(fn, world):
        out = [None] * world
        def body(r):
            try:
                out[r] = fn(r)
            except BaseException as e:       # an RtError is a result here
                out[r] = e
        ts = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
        for t in ts: t.start()
        for t in ts: t.join(150)
        assert not any(t.is_alive() for t in ts), "a rank is still waiting: the exchange hangs"
        return out

    def raw_sharded(ctx, cam, p, tile_rows, frame):
        rays = C.c_uint64(0)
        rc = lib.rt_render_frame_sharded(ctx._ctx, C.byref(cam.block), C.byref(p), tile_rows, _ffi.ptr(frame), C.byref(rays))
        return rc, rays.value, (lib.rt_last_error(ctx._ctx) or b"").decode()

    sp = rt.scene_book1_final(1)
    cams = [rt.Camera.look_at(16.0 / 9.0, 200, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0)),     # H = 112: 14 tiles of 8 rows
            rt.Camera.look_at(16.0 / 9.0, 136, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0)),     # H = 76: last tile ragged
            rt.Camera.look_at(2.0, 16, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0))]              # H = 8: one tile, ranks without any
    for world in (2, 3):
        ctxs = [rt.HipWorld(0, sp) for _ in range(world)]
        cid = rt.HipWorld.comm_id()
        res = run_ranks(lambda r: ctxs[r].comm_init(cid, world, r), world)
        assert all(x is None for x in res), res
        sends0, recvs0 = fake.fake_rccl_calls(0), fake.fake_rccl_calls(1)
        n_frames = 0
        for cam in cams:
            for chunk, tile_rows in ((1, 8), (0, 8), (1, 16), (3, 24)):
                p = rt.DeviceGUI(samples_per_pixel=6, max_depth=20, seed=12, sample_chunk=chunk)
                ref = ctxs[0].render(cam, p)
                for again in range(2):                                   # the second call reuses every buffer
                    res = run_ranks(lambda r: ctxs[r].render_frame_sharded(cam, p, tile_rows=tile_rows), world)
                    assert all(isinstance(x, dict) for x in res), res
                    assert np.array_equal(res[0]["rgba"], ref["rgba"]), (world, cam.image_width, chunk, tile_rows)
                    assert sum(x["rays"] for x in res) == ref["rays"]
                    n_frames += 1
        # every frame moved one tile buffer per peer through ncclSend / ncclRecv
        assert fake.fake_rccl_calls(0) - sends0 == n_frames * (world - 1)
        assert fake.fake_rccl_calls(1) - recvs0 == n_frames * (world - 1)

        # a rank that fails locally before the exchange: rank 0 without a frame to write to.  It returns its own
        # error, every other rank RT_ERR_COMM, nobody waits, no tile travels -- and the next frame is fine.
        cam = cams[0]
        p = rt.DeviceGUI(samples_per_pixel=6, max_depth=20, seed=12, sample_chunk=1)
        ref = ctxs[0].render(cam, p)
        frames = [np.zeros((cam.image_height, cam.image_width, 4), np.uint8) for _ in range(world)]
        sends0 = fake.fake_rccl_calls(0)
        res = run_ranks(lambda r: raw_sharded(ctxs[r], cam, p, 8, None if r == 0 else frames[r]), world)
        assert res[0][0] == RT_ERR_INVALID and "frame_out" in res[0][2], res
        assert all(x[0] == RT_ERR_COMM and "another rank failed" in x[2] for x in res[1:]), res
        assert fake.fake_rccl_calls(0) == sends0
        res = run_ranks(lambda r: ctxs[r].render_frame_sharded(cam, p, tile_rows=8), world)
        assert np.array_equal(res[0]["rgba"], ref["rgba"])

        # a rank on which the exchange itself fails (its all-reduce returns an error): it gives up its communicator,
        # which releases the peers; every rank returns RT_ERR_COMM, and that rank's context refuses further frames
        bad = world - 1
        fake.fake_rccl_fail_allreduce(bad, 1)
        aborts0 = fake.fake_rccl_calls(3)
        res = run_ranks(lambda r: raw_sharded(ctxs[r], cam, p, 8, frames[r]), world)
        assert all(x[0] == RT_ERR_COMM for x in res), res
        assert "ncclAllReduce" in res[bad][2]
        assert fake.fake_rccl_calls(3) == aborts0 + world          # the failed rank, then each released peer
        rc, _, msg = raw_sharded(ctxs[bad], cam, p, 8, frames[bad])
        assert rc == RT_ERR_COMM and "given up" in msg, (rc, msg)
        for c in ctxs:
            c.close()
    print("sharded threads ok")
""")


def test_sharded_frame_with_two_and_three_ranks_on_one_gpu(tmp_path):
    assert os.path.exists(FAKE), "tests/native/libfake_rccl.so is not built (make -C tests/native)"
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, RT_RCCL_LIB=FAKE)
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    assert "sharded threads ok" in res.stdout


def test_command_line_renderer_with_three_ranks_on_one_gpu(tmp_path):
    """tools/rt_render.cpp --gpus 3 --devices 0,0,0: the C++ host mirror (include/rt_host.hpp) drives
    rt_comm_init / rt_render_frame_sharded from one std::thread per rank, all on GPU 0 through the stand-in transport;
    the PPM it writes is byte for byte the one the single-GPU run writes."""
    import ray_tracing_amd as rt
    exe = os.path.join(os.path.dirname(rt.__file__), "csrc", "rt_render")
    common = ["--scene", "book1", "--width", "200", "--spp", "6", "--depth", "20", "--seed", "12", "--chunk", "1"]
    one, three = tmp_path / "one.ppm", tmp_path / "three.ppm"
    res = subprocess.run([exe] + common + ["--out", str(one)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    env = dict(os.environ, RT_RCCL_LIB=FAKE)
    res = subprocess.run([exe] + common + ["--gpus", "3", "--devices", "0,0,0", "--frames", "2", "--out", str(three)],
                         capture_output=True, text=True, timeout=240, env=env)
    assert res.returncode == 0, (res.stdout, res.stderr)
    assert "on 3 GPUs" in res.stdout
    assert open(one, "rb").read() == open(three, "rb").read()
    res = subprocess.run([exe] + common + ["--gpus", "2", "--devices", "0,7", "--out", str(three)],
                         capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 2 and "GPU 7" in res.stderr
