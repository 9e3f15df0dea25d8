"""GPU tests at the benchmark's full size (80kScene* stand-in, 1920x1080), where running the whole oracle would
take too long for a test: determinism, row-band invariance (the multi-GPU sharding mechanism, exercised on one
GPU), ray accounting, and oracle parity on a sample of rows, stage by stage, with the GPU's own buffers as the
oracle's inputs."""
import numpy as np
import pytest

import harness as h

rt = h.rt
pytestmark = pytest.mark.gpu

BUFS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXS = ("DepthBuffer", "NormalsBuffer", "DebugOverlay", "RayHitPointsA", "RayHitPointsB", "Result", "AccumulatedResult")


def snapshot(drv):
    drv.ctx.sync()
    return {**{n: drv.buffer(n) for n in BUFS}, **{n: drv.texture(n) for n in TEXS}}


def assert_equal_snap(a, b, what):
    for k in a:
        assert np.array_equal(h.bits(a[k]), h.bits(b[k])), f"{what}: {k} differs"


def test_deterministic_and_band_invariant(gpu_ctx, scene_80k):
    """Two runs give identical bits, and rendering the frame as three row bands (stage by stage, as the multi-GPU
    path does, with band edges that are not multiples of 16) gives the same bits as rendering it whole."""
    cfg = rt.default_config()
    full = rt.Driver(gpu_ctx, cfg, scene_80k)
    for _ in range(3):
        full.render_frame()
    want = snapshot(full)
    full.close()

    again = rt.Driver(gpu_ctx, cfg, scene_80k)
    for _ in range(3):
        again.render_frame()
    assert_equal_snap(snapshot(again), want, "second run")
    again.close()

    banded = rt.Driver(gpu_ctx, cfg, scene_80k)
    bands = [(0, 344), (344, 720), (720, 1080)]
    try:
        for _ in range(3):
            banded.update_per_frame()
            for stage in (0, 1):
                for b in bands:
                    gpu_ctx.set_row_band(*b)
                    banded.run_stage(stage)
            for i in range(cfg.SpatialReuseIterations):
                for b in bands:
                    gpu_ctx.set_row_band(*b)
                    banded.run_spatial_pass(i)
            for b in bands:
                gpu_ctx.set_row_band(*b)
                banded.run_transfer()
            for b in bands:
                gpu_ctx.set_row_band(*b)
                banded.run_stage(3)
                banded.run_post_processing()
        gpu_ctx.set_row_band(0, 0)
        assert_equal_snap(snapshot(banded), want, "banded run")
    finally:
        gpu_ctx.set_row_band(0, 0)
        banded.close()


def test_4k_two_bands_equal_full_frame(gpu_ctx, scene_80k):
    """3840x2160 (2040 regions in the heaviest-first table): the frame rendered as two bands, through the driver's
    own frame loop restricted to a band at a time, equals the frame rendered whole."""
    cfg = rt.default_config()
    cfg.Resolution[0], cfg.Resolution[1] = 3840, 2160
    names_b, names_t = ("Candidates", "HitInfos"), ("Result", "AccumulatedResult", "DepthBuffer")

    def snap(drv):
        drv.ctx.sync()
        return {**{n: drv.buffer(n) for n in names_b}, **{n: drv.texture(n) for n in names_t}}

    full = rt.Driver(gpu_ctx, cfg, scene_80k)
    for _ in range(2):
        full.render_frame()
    want = snap(full)
    full.close()
    assert np.isfinite(want["DepthBuffer"]).any() and not np.isfinite(want["DepthBuffer"]).all()

    banded = rt.Driver(gpu_ctx, cfg, scene_80k)
    bands = [(0, 1000), (1000, 2160)]
    try:
        for _ in range(2):
            banded.update_per_frame()
            for step in ([lambda: banded.run_stage(0)], [lambda: banded.run_stage(1)],
                         *[[lambda i=i: banded.run_spatial_pass(i)] for i in range(cfg.SpatialReuseIterations)],
                         [banded.run_transfer], [lambda: banded.run_stage(3), banded.run_post_processing]):
                for b in bands:
                    gpu_ctx.set_row_band(*b)
                    for f in step:
                        f()
        gpu_ctx.set_row_band(0, 0)
        assert_equal_snap(snap(banded), want, "4K banded run")
    finally:
        gpu_ctx.set_row_band(0, 0)
        banded.close()


def test_ray_accounting(gpu_ctx, scene_80k):
    cfg = rt.default_config()
    drv = rt.Driver(gpu_ctx, cfg, scene_80k)
    gpu_ctx.enable_stats(1)
    try:
        drv.render_frame()
        gpu_ctx.sync()
        gpu_ctx.profile_reset()
        drv.render_frame()
        p = gpu_ctx.profile()
        depth = drv.texture("DepthBuffer")
        hits = int(np.isfinite(depth).sum())
        assert p["primary"]["rays"] == 1920 * 1080
        assert 0 < p["candidates"]["rays"] <= hits            # one shadow ray per hit pixel facing its light sample
        assert 0 < p["trace_rays"]["rays"] <= hits            # one light ray per pixel with a usable reservoir
        res = drv.texture("Result")
        sky = ~np.isfinite(depth)
        assert np.array_equal(res[sky][:, :3], np.broadcast_to(np.float32([0.5, 0.7, 1.0]), (int(sky.sum()), 3)))
        # event brackets restricted to one kernel: the others are still counted but not timed
        gpu_ctx.profile_reset()
        gpu_ctx.profile_enable(True, only="trace_rays")
        drv.render_frame()
        p = gpu_ctx.profile()
        gpu_ctx.profile_enable(False)
        assert p["trace_rays"]["ms"] > 0 and p["primary"]["ms"] == 0 and p["primary"]["launches"] == 1
        assert p["primary"]["rays"] == 1920 * 1080
    finally:
        gpu_ctx.enable_stats(0)
        drv.close()


@pytest.mark.parametrize("rows", [(496, 520), (0, 8), (1072, 1080)])
def test_oracle_parity_on_row_sample(gpu_ctx, scene_80k, rows):
    """Frames 0..2 at 1920x1080.  Before each stage the oracle's buffers are overwritten with the GPU's, then both
    run the stage and the sampled rows must agree bit for bit (the spatial passes read the GPU-produced
    neighbourhood, so every row of the sample is a full-size result)."""
    cfg = rt.default_config()
    drv = rt.Driver(gpu_ctx, cfg, scene_80k)
    orc = h.OracleRenderer(scene_80k.arrays(), cfg, records=False, row0=rows[0], row1=rows[1])
    W = cfg.Resolution[0]
    sl = slice(rows[0] * W, rows[1] * W)

    def load_gpu_state():
        s = snapshot(drv)
        for n in BUFS:
            np.copyto(orc.buf[n], s[n])
        for n in TEXS:
            np.copyto(orc.tex[n], s[n])

    def compare(what):
        s = snapshot(drv)
        for n in BUFS:
            assert np.array_equal(h.bits(s[n][sl]), h.bits(orc.buf[n][sl])), f"{what}: {n}"
        for n in TEXS:
            assert np.array_equal(h.bits(s[n][rows[0]:rows[1]]), h.bits(orc.tex[n][rows[0]:rows[1]])), f"{what}: {n}"

    try:
        for frame in range(3):
            drv.update_per_frame()
            orc.update_per_frame()
            for stage in (0, 1):
                load_gpu_state()
                drv.run_stage(stage)
                orc.run_stage(stage)
                compare(f"frame {frame} stage {stage}")
            cycle = False
            for i in range(cfg.SpatialReuseIterations):
                cycle = not cycle
                load_gpu_state()
                drv.run_spatial_pass(i)
                orc.spatial_pass(i, cycle)
                compare(f"frame {frame} spatial pass {i}")
            load_gpu_state()
            drv.run_transfer()
            orc.transfer_to_original()
            compare(f"frame {frame} transfer")
            load_gpu_state()
            drv.run_stage(3)
            orc.trace_rays()
            compare(f"frame {frame} trace rays")
            load_gpu_state()
            drv.run_post_processing()
            orc.accumulate()
            compare(f"frame {frame} accumulate")
    finally:
        drv.close()
