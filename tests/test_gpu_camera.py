"""GPU parity with a camera that moves between frames, and with the reference's 8-bit render targets.

Main.SetCameraData (Main.cs:184-204) pushes this frame's camera and the previous frame's every frame; TemporalReuse
(ReSTIR_BVH_RayTracer.compute:586-641) then takes one of three paths per hit pixel:
  * the reprojected uv differs from the pixel's by more than TemporalPrecisionThreshold -> copy and return (:600-604),
  * the pixel's hit point moved by more than PixelMovementThreshold between RayHitPointsA and B -> no merge (:617 false),
  * otherwise the merge with the history (:617 true).
With a static camera and no defocus jitter only the last one runs.  Here the camera moves on some frames (first path)
and the lens jitter (DefocusStrength) moves silhouette pixels' hit points on the others (second path); the oracle counts
how many pixels took each path and every buffer is compared bit for bit after every frame.
"""
import numpy as np
import pytest

import harness as h
from test_gpu_parity import attach_records, compare_all

rt = h.rt
pytestmark = pytest.mark.gpu


quat = h.quat


@pytest.mark.parametrize("unorm8", [0, 1])
def test_moving_camera_takes_every_temporal_path(gpu_ctx, scene_small, unorm8):
    cfg = h.small_config(96, 64)
    cfg.DefocusStrength = 0.002      # moves a hit point 10 units away by about 0.02 from frame to frame
    cfg.PixelMovementThreshold = 0.02
    cfg.Unorm8RenderTargets = unorm8
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    rec = attach_records(gpu_ctx, drv, 96 * 64)
    pos0, rot0 = [float(x) for x in cfg.cameraPosition], [float(x) for x in cfg.cameraRotation]
    moves = {2: ([pos0[0] + 0.3, pos0[1] + 0.1, pos0[2]], rot0),                       # translation only
             4: ([pos0[0] + 0.3, pos0[1] + 0.1, pos0[2]], quat((0, 1, 0), 2.0, rot0)),  # rotation only
             5: ([pos0[0] - 0.2, pos0[1], pos0[2] + 0.4], quat((1, 0, 0), -1.5, rot0))}  # both, two frames in a row
    seen = {"temporalCopies": 0, "temporalMerges": 0, "temporalRejects": 0}
    per_frame = []
    try:
        for frame in range(8):
            if frame in moves:
                drv.set_camera(*moves[frame])
                orc.set_camera(*moves[frame])
            drv.render_frame()
            orc.render_frame()
            compare_all(gpu_ctx, drv, orc, rec, what=f"unorm8={unorm8} frame {frame}")
            c = orc.take_counters()
            per_frame.append({**{k: c[k] for k in seen}, "hit": int(np.isfinite(orc.tex["DepthBuffer"]).sum())})
            for k in seen:
                seen[k] += c[k]
        assert min(f["hit"] for f in per_frame) > 500, "scene not in view"
        # frames on which the camera moved copy (nearly) every hit pixel; the frames after them merge or reject
        for f in moves:
            assert per_frame[f]["temporalCopies"] > 0.9 * per_frame[f]["hit"], (f, per_frame[f])
        assert per_frame[3]["temporalCopies"] == 0 and per_frame[3]["temporalMerges"] > 0
        assert all(v > 0 for v in seen.values()), seen
        if unorm8:
            for name in ("RayHitPointsA", "RayHitPointsB", "NormalsBuffer", "Result", "AccumulatedResult", "DebugOverlay"):
                t = drv.texture(name)
                assert t.min() >= 0 and t.max() <= 1, name
                assert np.array_equal(np.round(t * 255) / np.float32(255), t), f"{name} is not a multiple of 1/255"
            assert drv.texture("Result").max() == 1.0   # radiance above 1 saturates in an 8-bit target
    finally:
        drv.close()
        gpu_ctx.lib.rt_buffer_release(rec)


def test_unorm8_targets_static_camera_full_sequence(gpu_ctx, scene_small):
    """The reference's own configuration of the render targets (8 bit) over four frames of the default sequence."""
    cfg = h.small_config(72, 56)
    cfg.Unorm8RenderTargets = 1
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    try:
        for frame in range(4):
            drv.render_frame()
            orc.render_frame()
            compare_all(gpu_ctx, drv, orc, what=f"unorm8 frame {frame}")
    finally:
        drv.close()


def test_ris_unorm8_targets(gpu_ctx, scene_small):
    """The RIS project's targets are all `new RenderTexture(w, h, 24)` (BVH_RIS_DL Main.cs:324-364): 8 bit."""
    cfg = h.small_ris_config(72, 40)
    cfg.Unorm8RenderTargets = 1
    drv = rt.RisDriver(gpu_ctx, cfg, scene_small)
    orc = h.RisOracleRenderer(scene_small.arrays(), cfg)
    try:
        for frame in range(2):
            drv.render_frame()
            orc.render_frame()
            gpu_ctx.sync()
            for name in ("Result", "AccumulatedResult", "DebugOverlay", "RayHitPoints"):
                assert np.array_equal(h.bits(drv.texture(name)), h.bits(orc.tex[name])), f"frame {frame} {name}"
            assert np.array_equal(h.bits(drv.buffer("HitInfos")), h.bits(orc.buf["HitInfos"]))
    finally:
        drv.close()
