"""Rehearsal of the multi-process GPU path on ONE GPU: two ranks share cuda:0, each renders its row band through
the C-ABI, halos and the frame buffer travel over gloo (staged through host memory, because RCCL refuses two ranks
on one device).  Rank 0 checks the gathered frame against a single-process full-frame render, bit for bit."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, res, frames, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import harness as h
    rt = h.rt
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    scene = rt.build_scene_s0(subdivide=False, threads=2)
    cfg = h.small_config(*res)
    runner = mg.DistributedRunner(rt, scene, cfg, rank, world, 0, backend="gloo")
    try:
        ref_ctx = ref = None
        if rank == 0:
            ref_ctx = rt.RtContext(0)
            ref = rt.Driver(ref_ctx, cfg, scene)
        for frame in range(frames):
            runner.render_frame()
            runner.sync()
            if rank == 0:
                ref.render_frame()
                ref_ctx.sync()
                assert np.array_equal(h.bits(runner.drv.texture("Result")), h.bits(ref.texture("Result"))), frame
                b1 = runner.bounds[1]
                assert np.array_equal(h.bits(runner.drv.texture("AccumulatedResult")[:b1]),
                                      h.bits(ref.texture("AccumulatedResult")[:b1])), frame
                lo, hi = 0, b1 * res[0]
                for name in ("Candidates", "TemporalFrameBuffer", "HitInfos"):
                    assert np.array_equal(h.bits(runner.drv.buffer(name)[lo:hi]), h.bits(ref.buffer(name)[lo:hi])), (frame, name)
        runner.barrier()
        if rank == 0:
            ref.close()
            ref_ctx.close()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        runner.close()


def test_two_ranks_one_gpu(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, (256, 192), 3, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_native_frame_loop_single_rank(gpu_ctx, scene_small):
    """The C++ multi-GPU frame loop with one rank (and a one-rank RCCL communicator, which exercises the RCCL set-up
    and tear-down on this GPU): must equal the plain driver bit for bit."""
    import harness as h
    rt = h.rt
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    cfg = h.small_config(128, 96)
    ref = rt.Driver(gpu_ctx, cfg, scene_small)
    for _ in range(3):
        ref.render_frame()
    want = {n: ref.texture(n) for n in ("Result", "AccumulatedResult")}
    want_c = ref.buffer("Candidates")
    ref.close()
    runner = mg.RcclRunner(rt, scene_small, cfg, 0, 1, 0, force_comm=True)
    try:
        assert runner.bounds == [0, 96]
        runner.render_frames(2)
        runner.render_frame()
        runner.sync()
        for n in want:
            assert np.array_equal(h.bits(runner.drv.texture(n)), h.bits(want[n])), n
        assert np.array_equal(h.bits(runner.drv.buffer("Candidates")), h.bits(want_c))
        # the halo transfer path itself, looped back to this rank: rows 40..52 -> rows 8..20
        depth = runner.drv.texture("DepthBuffer")
        assert runner.host.rth_mg_loopback(runner.h, 40, 52, 8) == 0, runner.host.rth_mg_last_error(runner.h)
        runner.sync()
        got_c, got_d = runner.drv.buffer("Candidates"), runner.drv.texture("DepthBuffer")
        W = 128
        assert np.array_equal(h.bits(got_c[8 * W:20 * W]), h.bits(want_c[40 * W:52 * W]))
        assert np.array_equal(h.bits(got_c[20 * W:]), h.bits(want_c[20 * W:])) and np.array_equal(h.bits(got_c[:8 * W]), h.bits(want_c[:8 * W]))
        assert np.array_equal(h.bits(got_d[8:20]), h.bits(depth[40:52])) and np.array_equal(h.bits(got_d[20:]), h.bits(depth[20:]))
        assert np.isfinite(depth[40:52]).any()
    finally:
        runner.close()
