"""The measured alternative "top of the tree in LDS" (rt_set_tuning "top_nodes_lds"): the merged candidate and TraceRays
launches take the first internal nodes - breadth first from the scene root into the objects' mesh BVHs - from a table
every wave loads into LDS.  Same nodes, same order, same arithmetic: every buffer and render target must equal the
oracle's bit for bit, whatever the number of slots, on a scene whose table holds scene-level nodes and several meshes
as well as on the two-object scene.
"""
import numpy as np
import pytest

import harness as h

rt = h.rt
pytestmark = pytest.mark.gpu

BUFFERS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXTURES = ("DepthBuffer", "NormalsBuffer", "DebugOverlay", "RayHitPointsA", "RayHitPointsB", "Result", "AccumulatedResult")


def frames_against_oracle(ctx, scene, cfg, frames=3):
    drv = rt.Driver(ctx, cfg, scene)
    orc = h.OracleRenderer(scene.arrays(), cfg)
    try:
        for f in range(frames):
            drv.render_frame()      # whole frames: the merged launches, which are the ones that use the table
            orc.render_frame()
            ctx.sync()
            for name in BUFFERS:
                assert np.array_equal(h.bits(drv.buffer(name)), h.bits(orc.buf[name])), f"frame {f}: {name} differs from the oracle"
            for name in TEXTURES:
                assert np.array_equal(h.bits(drv.texture(name)), h.bits(orc.tex[name])), f"frame {f}: {name} differs from the oracle"
        launches = {k: v["launches"] for k, v in ctx.profile().items()} if hasattr(ctx, "profile") else {}
        return orc, launches
    finally:
        drv.close()


@pytest.mark.parametrize("slots", [1, 5, 16, 64])
def test_two_object_scene(scene_small, slots):
    ctx = rt.RtContext(0)
    try:
        ctx.set_tuning("top_nodes_lds", slots)
        orc, _ = frames_against_oracle(ctx, scene_small, h.small_config(200, 120))
        assert np.isfinite(orc.tex["DepthBuffer"]).sum() > 0.1 * orc.tex["DepthBuffer"].size
    finally:
        ctx.close()


@pytest.mark.parametrize("slots", [7, 64])
@pytest.mark.parametrize("bounces", [1, 2])
def test_scene_bvh_of_several_levels_and_instanced_meshes(slots, bounces):
    scene = rt.build_scene_instanced(grid=(4, 3), emissive_objects=3, emissive_tris=24, unique=False, subdivide=False)
    cfg = h.small_config(96, 64)
    cfg.cameraPosition[0], cfg.cameraPosition[1], cfg.cameraPosition[2] = -4.0, 6.0, -25.0
    cfg.cameraRotation[0], cfg.cameraRotation[1], cfg.cameraRotation[2], cfg.cameraRotation[3] = 0.0, 0.0, 0.0, 1.0
    cfg.MaxBounceCount = bounces
    ctx = rt.RtContext(0)
    try:
        ctx.set_tuning("top_nodes_lds", slots)
        orc, _ = frames_against_oracle(ctx, scene, cfg)
        assert np.isfinite(orc.tex["DepthBuffer"]).sum() > 200
    finally:
        ctx.close()


def test_switching_the_table_on_and_off_between_frames(scene_small):
    """The prepared scene is rebuilt when the setting changes; frames before and after are the oracle's."""
    ctx = rt.RtContext(0)
    cfg = h.small_config(104, 72)
    drv = rt.Driver(ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    try:
        for f, slots in enumerate([0, 32, 32, 0, 64, 3]):
            ctx.set_tuning("top_nodes_lds", slots)
            drv.render_frame()
            orc.render_frame()
            ctx.sync()
            for name in ("Candidates", "TemporalFrameBuffer"):
                assert np.array_equal(h.bits(drv.buffer(name)), h.bits(orc.buf[name])), f"frame {f} ({slots} slots): {name}"
            assert np.array_equal(h.bits(drv.texture("Result")), h.bits(orc.tex["Result"])), f"frame {f} ({slots} slots): Result"
    finally:
        drv.close()
        ctx.close()


def test_the_merged_launches_read_the_table(scene_small):
    """The log of launches (rt_debug_trace) lists the table among the objects a candidate / TraceRays launch reads when -
    and only when - the launch is the variant that keeps it in LDS."""
    def objects_per_launch(slots):
        ctx = rt.RtContext(0)
        drv = rt.Driver(ctx, h.small_config(64, 64), scene_small)
        try:
            ctx.set_tuning("top_nodes_lds", slots)
            drv.render_frame()
            ctx.sync()
            ctx.trace(True)
            drv.render_frame()
            ctx.sync()
            ctx.trace(False)
            lines = [l.split() for l in ctx.trace_text().splitlines() if l.startswith("L ")]
            return {k: [len(l) - 3 for l in lines if l[2] == k] for k in ("candidates", "trace_rays", "primary", "spatial")}
        finally:
            drv.close()
            ctx.close()
    without, with_table = objects_per_launch(0), objects_per_launch(48)
    for k in ("candidates", "trace_rays"):
        assert len(without[k]) == 1 and len(with_table[k]) == 1
        assert with_table[k][0] == without[k][0] + 1, (k, without[k], with_table[k])
    for k in ("primary", "spatial"):
        assert with_table[k] == without[k]


def test_argument_is_checked(gpu_ctx):
    for bad in (-1, 65):
        with pytest.raises(rt.NativeError):
            gpu_ctx.set_tuning("top_nodes_lds", bad)
