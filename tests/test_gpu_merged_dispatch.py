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


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_calls_between_dispatches(gpu_ctx, scene_small, seed):
    """Whatever is called between the dispatches of a frame - nothing, a sync, a partial readback, a band set to the same
    rows, a statistics query - the frames must come out as when every dispatch is waited for."""
    rng = np.random.default_rng(seed)
    cfg = h.small_config(88, 64)
    cfg.SpatialReuseIterations = int(rng.integers(1, 4))      # odd counts end with TransferToOriginal, even ones do not
    loose, strict = rt.Driver(gpu_ctx, cfg, scene_small), rt.Driver(gpu_ctx, cfg, scene_small)

    def between(drv):
        k = int(rng.integers(0, 6))
        if k == 1:
            gpu_ctx.sync()
        elif k == 2:
            drv.buffer_rows("Candidates", 8, 16)
        elif k == 3:
            drv.texture_rows("Result", 0, 8)
        elif k == 4:
            gpu_ctx.set_row_band(0, 0)
        elif k == 5:
            gpu_ctx.profile()

    def frame(drv, hook):
        drv.update_per_frame(); hook(drv)
        drv.run_stage(0); hook(drv)
        drv.run_stage(1); hook(drv)
        for i in range(cfg.SpatialReuseIterations):
            drv.run_spatial_pass(i); hook(drv)
        if cfg.SpatialReuseIterations % 2 == 1:
            drv.run_transfer(); hook(drv)
        drv.run_stage(3); hook(drv)
        drv.run_post_processing(); hook(drv)

    try:
        for _ in range(5):
            frame(loose, between)
        for _ in range(5):
            frame(strict, lambda d: gpu_ctx.sync())
        gpu_ctx.sync()
        for k in BUFS:
            assert np.array_equal(h.bits(loose.buffer(k)), h.bits(strict.buffer(k))), k
        for k in TEXS:
            assert np.array_equal(h.bits(loose.texture(k)), h.bits(strict.texture(k))), k
    finally:
        gpu_ctx.set_row_band(0, 0)
        loose.close()
        strict.close()
