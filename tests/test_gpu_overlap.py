"""The next frame's primary rays beside the previous frame's last kernels (rt_set_overlap).

The library alternates between two copies of HitInfos and DepthBuffer so that InitialTrace's primary rays need not wait
for the SpatialReusePass / TraceRays launches of the frame before.  Whatever a host reads, whenever it reads it, must be
what the dispatches left there in call order: frames enqueued back to back equal frames rendered one launch at a time,
and HitInfos / DepthBuffer read between any two dispatches are the current frame's.
"""
import numpy as np
import pytest

import harness as h

rt = h.rt
pytestmark = pytest.mark.gpu

BUFS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXS = ("Result", "AccumulatedResult", "DepthBuffer", "NormalsBuffer", "DebugOverlay", "RayHitPointsA", "RayHitPointsB")


def snapshot(drv):
    drv.ctx.sync()
    return {**{k: drv.buffer(k) for k in BUFS}, **{k: drv.texture(k) for k in TEXS}}


@pytest.mark.parametrize("unorm8", [0, 1])
def test_overlapped_frames_equal_serial_frames_1080p(gpu_ctx, scene_80k, unorm8):
    cfg = rt.default_config()
    cfg.Unorm8RenderTargets = unorm8
    frames = 7   # odd: the last frame leaves the buffers on their second copies
    try:
        gpu_ctx.set_overlap(0)
        serial = rt.Driver(gpu_ctx, cfg, scene_80k)
        for _ in range(frames):
            serial.render_frame()
        want = snapshot(serial)
        serial.close()
        gpu_ctx.set_overlap(1)
        beside = rt.Driver(gpu_ctx, cfg, scene_80k)
        for _ in range(frames):
            beside.render_frame()          # no call in between: frame f+1's primary rays meet frame f's kernels
        got = snapshot(beside)
        beside.close()
        for k in want:
            assert np.array_equal(h.bits(want[k]), h.bits(got[k])), k
        assert np.isfinite(got["DepthBuffer"]).sum() > 0.1 * got["DepthBuffer"].size
    finally:
        gpu_ctx.set_overlap(1)


def test_reads_between_the_dispatches_of_consecutive_frames(gpu_ctx, scene_small):
    cfg = h.small_config(104, 72)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)

    def check(what):
        for k in ("HitInfos",):
            assert np.array_equal(h.bits(drv.buffer(k)), h.bits(orc.buf[k])), f"{what}: {k}"
        for k in ("DepthBuffer", "DebugOverlay", "NormalsBuffer", "RayHitPointsA", "RayHitPointsB"):
            assert np.array_equal(h.bits(drv.texture(k)), h.bits(orc.tex[k])), f"{what}: {k}"

    try:
        for f in range(4):
            drv.update_per_frame(); orc.update_per_frame()
            check(f"frame {f} before InitialTrace")          # still the previous frame's
            drv.run_stage(0); orc.initial_trace()
            check(f"frame {f} after InitialTrace")
            drv.run_stage(1); orc.temporal_reuse()
            check(f"frame {f} after TemporalReuse")
            cycle = False
            for i in range(cfg.SpatialReuseIterations):
                cycle = not cycle
                drv.run_spatial_pass(i); orc.spatial_pass(i, cycle)
                check(f"frame {f} after spatial pass {i}")
            drv.run_transfer(); orc.transfer_to_original()
            drv.run_stage(3); orc.trace_rays()
            check(f"frame {f} after TraceRays")
            drv.run_post_processing(); orc.accumulate()
        for k in BUFS:
            assert np.array_equal(h.bits(drv.buffer(k)), h.bits(orc.buf[k])), k
        for k in TEXS:
            assert np.array_equal(h.bits(drv.texture(k)), h.bits(orc.tex[k])), k
    finally:
        drv.close()


def test_unsynchronised_frames_against_the_oracle(gpu_ctx, scene_small):
    """Five frames enqueued without a single wait, then everything against the oracle; a moving camera makes the
    TemporalReuse of frame f read both hit-point targets, one of which frame f+1's primary rays overwrite."""
    cfg = h.small_config(200, 120)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    try:
        pos = np.array(cfg.cameraPosition[:], np.float32)
        rot = [float(x) for x in cfg.cameraRotation]
        for f in range(5):
            p = pos + np.float32(0.002 * f) * np.array([1, 0, 0.5], np.float32)
            drv.set_camera(p, rot); orc.set_camera(p, rot)
            drv.render_frame(); orc.render_frame()
        for k in BUFS:
            assert np.array_equal(h.bits(drv.buffer(k)), h.bits(orc.buf[k])), k
        for k in TEXS:
            assert np.array_equal(h.bits(drv.texture(k)), h.bits(orc.tex[k])), k
    finally:
        drv.close()


def test_ris_frames_overlap_and_match(gpu_ctx, scene_small):
    cfg = h.small_ris_config(120, 88)
    drv = rt.RisDriver(gpu_ctx, cfg, scene_small)
    orc = h.RisOracleRenderer(scene_small.arrays(), cfg)
    try:
        for f in range(4):
            drv.update_per_frame(); orc.update_per_frame()
            drv.run_stage(0); orc.generate_candidates()
            assert np.array_equal(h.bits(drv.buffer("HitInfos")), h.bits(orc.buf["HitInfos"])), f"frame {f}"
            drv.run_stage(1); orc.trace_rays()
            assert np.array_equal(h.bits(drv.buffer("HitInfos")), h.bits(orc.buf["HitInfos"])), f"frame {f}"
            drv.run_post_processing(); orc.accumulate()
        for _ in range(3):
            drv.render_frame(); orc.render_frame()
        for k in ("HitInfos", "Candidates"):
            assert np.array_equal(h.bits(drv.buffer(k)), h.bits(orc.buf[k])), k
        for k in ("Result", "AccumulatedResult", "DebugOverlay", "RayHitPoints"):
            assert np.array_equal(h.bits(drv.texture(k)), h.bits(orc.tex[k])), k
    finally:
        drv.close()


def test_banded_dispatches_of_one_frame_do_not_flip(gpu_ctx, scene_small):
    """Default mode: an InitialTrace on a row band leaves the copies alone, so three bands rendered one after the other
    into the same buffers still add up to the frame."""
    cfg = h.small_config(96, 96)
    full, banded = rt.Driver(gpu_ctx, cfg, scene_small), rt.Driver(gpu_ctx, cfg, scene_small)
    try:
        for _ in range(2):
            full.render_frame()
            banded.update_per_frame()
            for b in ((0, 32), (32, 72), (72, 96)):
                gpu_ctx.set_row_band(*b)
                banded.run_stage(0)
            gpu_ctx.set_row_band(0, 0)
            for st in (1, 2, 3):
                banded.run_stage(st)
            banded.run_post_processing()
        want, got = snapshot(full), snapshot(banded)
        for k in want:
            assert np.array_equal(h.bits(want[k]), h.bits(got[k])), k
    finally:
        gpu_ctx.set_row_band(0, 0)
        full.close()
        banded.close()


@pytest.mark.parametrize("name,value", [("spatial_lds_max_offset", 3), ("spatial_lds_max_offset", 1), ("primary_background_groups", 64),
                                        ("overlap_behind_candidates", 0)])
def test_speed_only_choices_leave_every_bit_alone(gpu_ctx, scene_small, name, value):
    """rt_set_tuning: the alternatives that were measured and not adopted still render the same frames - the LDS-staged
    SpatialReusePass (also with a window of +-1, and with blocks at x < MaxOffset whose neighbours wrap into the row
    above), primary rays as a background launch, primary rays started as early as their inputs allow."""
    cfg = h.small_config(488, 264)   # 31 x 17 blocks, ragged on both edges
    cfg.SpatialReuseBlur = 1 if (name, value) == ("spatial_lds_max_offset", 1) else 3
    try:
        plain = rt.Driver(gpu_ctx, cfg, scene_small)
        for _ in range(4):
            plain.render_frame()
        want = snapshot(plain)
        plain.close()
        gpu_ctx.set_tuning(name, value)
        tuned = rt.Driver(gpu_ctx, cfg, scene_small)
        for _ in range(4):
            tuned.render_frame()
        got = snapshot(tuned)
        tuned.close()
        for k in want:
            assert np.array_equal(h.bits(want[k]), h.bits(got[k])), k
        with pytest.raises(rt.NativeError):
            gpu_ctx.set_tuning("no_such_choice", 1)
    finally:
        for n, v in (("spatial_lds_max_offset", 0), ("primary_background_groups", 0), ("overlap_behind_candidates", 1)):
            gpu_ctx.set_tuning(n, v)
