"""InitialTrace's candidate kernel + TemporalReuse, and TransferToOriginal + TraceRays + AccumulateFrames, issued back
to back run as one launch each (the library holds a dispatch until the next call shows whether the two can be merged);
anything in between - here a sync - makes them run one by one.  Both orders of events must leave every buffer and render target identical, and the launch counts must show
which path ran."""
import numpy as np
import pytest

import harness as h

rt = h.rt
pytestmark = pytest.mark.gpu

BUFS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXS = ("Result", "AccumulatedResult", "DepthBuffer", "RayHitPointsA", "RayHitPointsB")


def launches(ctx):
    return {k: v["launches"] for k, v in ctx.profile().items()}


@pytest.mark.parametrize("unorm8", [0, 1])
def test_merged_and_separate_launches_agree(gpu_ctx, scene_small, unorm8):
    cfg = h.small_config(104, 72)
    cfg.Unorm8RenderTargets = unorm8
    merged, apart = rt.Driver(gpu_ctx, cfg, scene_small), rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    frames = 3
    try:
        gpu_ctx.profile_reset()
        gpu_ctx.profile_enable(True)
        for _ in range(frames):
            merged.render_frame()
        gpu_ctx.sync()
        n = launches(gpu_ctx)
        assert n.get("trace_rays", 0) == frames and n.get("transfer", 0) == 0 and n.get("accumulate", 0) == 0, n
        # (the first TemporalReuse of a context converts TemporalFrameBuffer to the library's layout, which launches
        # what is held: at most that one runs on its own)
        assert n.get("candidates", 0) == frames and n.get("temporal", 0) <= 1, n
        got = {**{k: merged.buffer(k) for k in BUFS}, **{k: merged.texture(k) for k in TEXS}}

        gpu_ctx.profile_reset()
        for _ in range(frames):
            apart.update_per_frame()
            for step in ([lambda: apart.run_stage(0)], [lambda: apart.run_stage(1)],
                         *[[lambda i=i: apart.run_spatial_pass(i)] for i in range(cfg.SpatialReuseIterations)],
                         [apart.run_transfer], [lambda: apart.run_stage(3)], [apart.run_post_processing]):
                step[0]()
                gpu_ctx.sync()
        n = launches(gpu_ctx)
        assert n.get("trace_rays", 0) == frames and n.get("transfer", 0) == frames and n.get("accumulate", 0) == frames, n
        assert n.get("candidates", 0) == frames and n.get("temporal", 0) == frames, n
        gpu_ctx.profile_enable(False)
        for k in BUFS:
            assert np.array_equal(h.bits(got[k]), h.bits(apart.buffer(k))), k
        for k in TEXS:
            assert np.array_equal(h.bits(got[k]), h.bits(apart.texture(k))), k
        for _ in range(frames):
            orc.render_frame()
        assert np.array_equal(h.bits(got["Result"]), h.bits(orc.tex["Result"]))
        assert np.array_equal(h.bits(got["AccumulatedResult"]), h.bits(orc.tex["AccumulatedResult"]))
        assert np.array_equal(h.bits(got["Candidates"]), h.bits(orc.buf["Candidates"]))
    finally:
        merged.close()
        apart.close()


def test_a_readback_between_the_dispatches_sees_the_copy(gpu_ctx, scene_small):
    """TransferToOriginal is held back when it is dispatched; reading Candidates right after it must show its effect."""
    cfg = h.small_config(64, 48)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    try:
        for _ in range(2):
            drv.update_per_frame(); orc.update_per_frame()
            drv.run_stage(0); orc.run_stage(0)
            drv.run_stage(1); orc.run_stage(1)
            cycle = False
            for i in range(cfg.SpatialReuseIterations):
                cycle = not cycle
                drv.run_spatial_pass(i); orc.spatial_pass(i, cycle)
            drv.run_transfer(); orc.transfer_to_original()
            assert np.array_equal(h.bits(drv.buffer("Candidates")), h.bits(orc.buf["Candidates"]))   # no sync in between
            drv.run_stage(3); orc.trace_rays()
            assert np.array_equal(h.bits(drv.texture("Result")), h.bits(orc.tex["Result"]))
            drv.run_post_processing(); orc.accumulate()
            assert np.array_equal(h.bits(drv.texture("AccumulatedResult")), h.bits(orc.tex["AccumulatedResult"]))
    finally:
        drv.close()
