"""Edge cases of the path, GPU vs oracle bit for bit, and the error behaviour of the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import harness as h

rt = h.rt
pytestmark = pytest.mark.gpu

BUFS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXS = ("DepthBuffer", "NormalsBuffer", "DebugOverlay", "RayHitPointsA", "RayHitPointsB", "Result", "AccumulatedResult")


def run_and_compare(ctx, scene, cfg, frames=3):
    drv = rt.Driver(ctx, cfg, scene)
    orc = h.OracleRenderer(scene.arrays(), cfg, records=False)
    try:
        for f in range(frames):
            drv.render_frame()
            orc.render_frame()
            ctx.sync()
            for n in BUFS:
                assert np.array_equal(h.bits(drv.buffer(n)), h.bits(orc.buf[n])), (f, n)
            for n in TEXS:
                assert np.array_equal(h.bits(drv.texture(n)), h.bits(orc.tex[n])), (f, n)
        return orc
    finally:
        drv.close()


@pytest.mark.parametrize("res", [(1, 1), (8, 8), (9, 7), (17, 33), (130, 5)])
def test_tiny_and_ragged_resolutions(gpu_ctx, scene_small, res):
    run_and_compare(gpu_ctx, scene_small, h.small_config(*res))


def test_everything_misses(gpu_ctx, scene_small):
    """Camera turned away from the scene: every primary ray misses, no reservoir is ever written."""
    cfg = h.small_config(64, 40)
    cfg.cameraRotation[0], cfg.cameraRotation[1], cfg.cameraRotation[2], cfg.cameraRotation[3] = 0.0, 0.8571673, 0.0, 0.5150381
    orc = run_and_compare(gpu_ctx, scene_small, cfg)
    assert not np.isfinite(orc.tex["DepthBuffer"]).any()
    assert not orc.buf["Candidates"].view(np.uint8).any()


@pytest.mark.parametrize("iters", [0, 1, 2, 4])
def test_spatial_iteration_counts(gpu_ctx, scene_small, iters):
    """Odd counts end in CandidatesB and need TransferToOriginal, even counts do not (Main.cs:387-402)."""
    cfg = h.small_config(72, 56)
    cfg.SpatialReuseIterations = iters
    run_and_compare(gpu_ctx, scene_small, cfg)


@pytest.mark.parametrize("bounces", [0, 2, 3])
def test_bounce_counts(gpu_ctx, scene_small, bounces):
    """MaxBounceCount other than the designed 1: no light ray at all (0), or further bounces that keep the stale
    inverse direction of the first (2, 3) exactly as the reference's Trace loop does."""
    cfg = h.small_config(64, 48)
    cfg.MaxBounceCount = bounces
    run_and_compare(gpu_ctx, scene_small, cfg)


def test_no_temporal_reuse_and_many_candidates(gpu_ctx, scene_small):
    cfg = h.small_config(64, 48)
    cfg.TemporalReuseWeight = 0.0
    cfg.TriReservoirTestsNum = 32          # the "32 candidates" configuration
    cfg.SceneObjectReservoirTestsNum = 5
    cfg.TemporalCandidatesNum = 3
    cfg.SpatialReuseBlur = 40              # neighbourhood larger than the image: most offsets fall outside
    run_and_compare(gpu_ctx, scene_small, cfg)


def test_tight_thresholds_reject_neighbours(gpu_ctx, scene_small):
    cfg = h.small_config(96, 64)
    cfg.SpatialHitPointDiffThreshold = 0.05
    cfg.SpatialNormalsAngleThreshold = 10.0
    cfg.SpatialBRDFThreshold = 0.5
    cfg.PixelMovementThreshold = 0.0
    run_and_compare(gpu_ctx, scene_small, cfg)


def test_defocus_and_moved_camera(gpu_ctx, scene_small):
    cfg = h.small_config(64, 64)
    cfg.DefocusStrength = 0.05
    cfg.FocalPlaneFactor = 6.0
    cfg.cameraPosition[0], cfg.cameraPosition[1], cfg.cameraPosition[2] = 6.0, 4.0, 3.0
    run_and_compare(gpu_ctx, scene_small, cfg)


def test_many_objects_deep_scene_bvh(gpu_ctx):
    """A scene BVH with several levels, instanced meshes and several lights."""
    scene = rt.build_scene_instanced(grid=(4, 3), emissive_objects=3, emissive_tris=24, unique=False, subdivide=False)
    cfg = h.small_config(96, 64)
    cfg.cameraPosition[0], cfg.cameraPosition[1], cfg.cameraPosition[2] = -4.0, 6.0, -25.0
    cfg.cameraRotation[0], cfg.cameraRotation[1], cfg.cameraRotation[2], cfg.cameraRotation[3] = 0.0, 0.0, 0.0, 1.0
    orc = run_and_compare(gpu_ctx, scene, cfg)
    assert np.isfinite(orc.tex["DepthBuffer"]).sum() > 200


# ---------------------------------------------------------------- error behaviour of the C-ABI
def test_unbound_and_invalid_arguments(gpu_ctx):
    lib = gpu_ctx.lib
    sh = gpu_ctx.open_shader("ReSTIR_BVH_RayTracer")
    try:
        assert lib.rt_find_kernel(sh, b"InitialTrace") == 0 and lib.rt_find_kernel(sh, b"TraceRays") == 4
        assert lib.rt_find_kernel(sh, b"NoSuchKernel") < 0
        assert lib.rt_dispatch(sh, 0, 8, 8, 1) == -4            # RT_ERR_UNBOUND: Resolution not set
        res = (C.c_int * 2)(64, 64)
        lib.rt_set_ints(sh, b"Resolution", res, 2)
        assert lib.rt_dispatch(sh, 0, 8, 8, 1) == -4            # scene buffers not bound
        assert b"BVs" in lib.rt_last_error(gpu_ctx.h)
        assert lib.rt_dispatch(sh, 0, 4, 8, 1) == -1            # groups do not cover the resolution
        assert lib.rt_dispatch(sh, 7, 8, 8, 1) < 0
        assert lib.rt_set_int(sh, b"SomethingTheShaderDoesNotDeclare", 3) == 0   # ignored, like Unity
        assert lib.rt_enable_keyword(sh, b"NOT_A_KEYWORD") == 0
        h_ = C.c_void_p()
        assert lib.rt_buffer_create(gpu_ctx.h, 0, 52, C.byref(h_)) < 0
        assert lib.rt_buffer_create(gpu_ctx.h, 4, 3, C.byref(h_)) < 0
        assert lib.rt_shader_open(gpu_ctx.h, b"Nope", C.byref(h_)) < 0
    finally:
        lib.rt_shader_close(sh)


def test_bad_scene_is_rejected(gpu_ctx, scene_small):
    """A BVH child index beyond the buffer must be refused on the host, never handed to a kernel."""
    a = scene_small.arrays()
    bad = a["BVs"].copy()
    root = a["sceneBVHStartIndex"]
    bad["indexB"][root] = -(len(bad) + 5)
    cfg = h.small_config(32, 32)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    try:
        gpu_ctx.set_buffer_data(drv.buffer_handle("BVs"), bad)
        with pytest.raises(rt.NativeError, match="BVH"):
            drv.render_frame()
        gpu_ctx.set_buffer_data(drv.buffer_handle("BVs"), a["BVs"])
        drv.render_frame()                                        # and recovers once the data is valid again
        gpu_ctx.sync()
    finally:
        drv.close()


def test_bad_triangles_and_shared_subtrees_are_rejected(gpu_ctx, scene_small):
    """The triangle indices are checked on the device and the mesh BVHs on several host threads: a vertex index beyond
    the vertex buffer, and a node that two parents name, are both refused."""
    a = scene_small.arrays()
    cfg = h.small_config(32, 32)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    try:
        tris = a["Triangles"].copy()
        tris["v"][len(tris) // 2][1] = len(a["Vertices"]) + 3
        gpu_ctx.set_buffer_data(drv.buffer_handle("Triangles"), tris)
        with pytest.raises(rt.NativeError, match="vertex index"):
            drv.render_frame()
        gpu_ctx.set_buffer_data(drv.buffer_handle("Triangles"), a["Triangles"])
        bvs = a["BVs"].copy()
        start = int(a["SceneObjects"]["bvStartIndex"].max())       # a mesh root: an internal node
        assert bvs["indexA"][start] < 0
        bvs["indexB"][start] = bvs["indexA"][start]                # both children the same node
        gpu_ctx.set_buffer_data(drv.buffer_handle("BVs"), bvs)
        with pytest.raises(rt.NativeError, match="reachable twice"):
            drv.render_frame()
        gpu_ctx.set_buffer_data(drv.buffer_handle("BVs"), a["BVs"])
        drv.render_frame()
        gpu_ctx.sync()
    finally:
        drv.close()


def test_stamps_and_row_reports_check_their_arguments(gpu_ctx):
    lib = gpu_ctx.lib
    assert lib.rt_wait_mark(gpu_ctx.h, 0, 63) < 0 and b"never recorded" in lib.rt_last_error(gpu_ctx.h)
    assert lib.rt_wait_mark(gpu_ctx.h, 2, 0) < 0 and lib.rt_wait_mark(gpu_ctx.h, 0, 64) < 0
    gpu_ctx.check(lib.rt_mark(gpu_ctx.h, 0, 40), "rt_mark")
    gpu_ctx.check(lib.rt_wait_mark(gpu_ctx.h, 0, 40), "rt_wait_mark")     # same stream: a no-op that must be accepted
    gpu_ctx.check(lib.rt_wait_mark(gpu_ctx.h, 1, 40), "rt_wait_mark")     # creates the transfer stream
    buf = gpu_ctx.create_buffer(64, 52)
    try:
        assert lib.rt_buffer_rows_written(buf, 8, 0, 9) < 0               # 9 rows of 8 do not fit 64 records
        assert lib.rt_buffer_rows_written(buf, 8, 5, 4) < 0
        gpu_ctx.check(lib.rt_buffer_rows_written(buf, 8, 0, 8), "rt_buffer_rows_written")   # a buffer with one copy: nothing to do
    finally:
        lib.rt_buffer_release(buf)
    for name, bad in (("frames_in_flight", 5), ("one_wave_groups", 32), ("one_wave_groups", -1), ("table_refresh_frames", 0),
                      ("region_blocks_rays", 3), ("region_blocks_primary", 16), ("region_blocks_reuse", 0)):
        with pytest.raises(rt.NativeError):
            gpu_ctx.set_tuning(name, bad)
    gpu_ctx.sync()


def test_buffer_roundtrip_keeps_abi_layout(gpu_ctx, scene_small):
    """Reservoir buffers live as planes inside the library; set_data/get_data always speak the 52-byte records."""
    cfg = h.small_config(16, 16)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    try:
        drv.render_frame()
        rng = np.random.default_rng(3)
        data = rng.standard_normal((256, 13)).astype(np.float32).view(rt.DTYPES["Reservoir"]).reshape(-1)
        gpu_ctx.set_buffer_data(drv.buffer_handle("CandidatesB"), data)
        assert np.array_equal(h.bits(drv.buffer("CandidatesB")), h.bits(data))
        drv.render_frame()                                        # converts back to planes on the next dispatch
        gpu_ctx.sync()
    finally:
        drv.close()
