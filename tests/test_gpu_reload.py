"""The reference's scene reload through the boundary (Main.cs:117-132, 206-214, 273-310; SceneObjectSettings.cs:27-36).

A moved scene object makes the next Update run UpdateSettings(true) AFTER UpdatePerFrame: all ten buffers are released
and created again (the per-pixel ones zeroed), the scene arrays are uploaded anew, CalcTriNormals runs again - while
the frame that is about to be rendered keeps the FrameCount already pushed (TemporalReuse merges against a zero
TemporalFrameBuffer with FrameCount != 0), the render targets keep their contents (RayHitPointsA/B, AccumulatedResult),
and the count restarts at 0 with the frame after.  GPU (C++ frame driver through the C-ABI) against the oracle's twin
of that sequence, every buffer and render target bit for bit after every frame.
"""
import ctypes as C

import numpy as np
import pytest

import harness as h

rt = h.rt
pytestmark = pytest.mark.gpu

BUFS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXS = ("Result", "AccumulatedResult", "DepthBuffer", "NormalsBuffer", "DebugOverlay", "RayHitPointsA", "RayHitPointsB")
RT_ERR_UNBOUND = -4
RT_FORMAT_RGBA32F = 2


def compare(drv, orc, what):
    for k in BUFS:
        assert np.array_equal(h.bits(drv.buffer(k)), h.bits(orc.buf[k])), f"{what}: {k} differs from the oracle"
    for k in TEXS:
        assert np.array_equal(h.bits(drv.texture(k)), h.bits(orc.tex[k])), f"{what}: {k} differs from the oracle"


def gpu_frame(drv, ctx, cfg, synced, post=True):
    """One frame dispatch by dispatch; synced: every dispatch is waited for (nothing is merged or held)."""
    wait = ctx.sync if synced else (lambda: None)
    drv.update_per_frame(); wait()
    drv.run_stage(0); wait()
    drv.run_stage(1); wait()
    for i in range(cfg.SpatialReuseIterations):
        drv.run_spatial_pass(i); wait()
    drv.run_transfer(); wait()
    drv.run_stage(3); wait()
    if post:
        drv.run_post_processing(); wait()


def orc_frame(orc, post=True):
    orc.update_per_frame()
    orc.initial_trace()
    orc.temporal_reuse()
    orc.spatial_reuse()
    orc.trace_rays()
    if post:
        orc.accumulate()


@pytest.mark.parametrize("synced", [False, True])
@pytest.mark.parametrize("unorm8", [0, 1])
def test_reload_after_an_object_moved(gpu_ctx, synced, unorm8):
    scene = rt.build_scene_s0(subdivide=False)   # a scene of its own: the test moves an object
    cfg = h.small_config(104, 72)
    cfg.Unorm8RenderTargets = unorm8
    drv = rt.Driver(gpu_ctx, cfg, scene)
    orc = h.OracleRenderer(scene.arrays(), cfg)
    try:
        for f in range(3):
            # frame 2 leaves out AccumulateFrames: on the merged path its TraceRays is then still HELD when the reload's
            # releases arrive, and must be launched before the buffers it reads are freed
            gpu_frame(drv, gpu_ctx, cfg, synced, post=f < 2)
            orc_frame(orc, post=f < 2)
        n_bvs = len(scene.arrays()["BVs"])
        # the dragon (second object registered) moves and turns
        scene.set_object_transform(1, (-3.0, -2.0, 12.0), (0.0, 25.0, 0.0), (12.0, 12.0, 12.0))
        scene.rebuild()
        a = scene.arrays()
        # the reference appends the new scene BVH behind the old one (ObjectHelper.cs:38,613-614)
        assert len(a["BVs"]) == n_bvs + 3 and a["sceneBVHStartIndex"] == n_bvs
        drv.reload()
        orc.reload(a)
        for f in range(3, 7):
            gpu_frame(drv, gpu_ctx, cfg, synced)
            orc_frame(orc)
            if f == 3:   # the frame that reloaded: rendered with the count pushed before UpdateSettings zeroed it
                assert orc.p.FrameCount == 3 and orc.FrameCount == 0
                assert len(drv.buffer("BVs")) == n_bvs + 3
                tri = drv.buffer("Triangles")
                assert np.array_equal(h.bits(tri), h.bits(orc.a["Triangles"])), "CalcTriNormals after the reload"
            compare(drv, orc, f"frame {f} ({'synced' if synced else 'merged'})")
        assert orc.p.FrameCount == 2   # frames 4, 5, 6 count 0, 1, 2
        hits = np.isfinite(orc.tex["DepthBuffer"]).sum()
        assert hits > 0.1 * orc.tex["DepthBuffer"].size
    finally:
        drv.close()


def test_reload_of_an_unchanged_scene_with_new_settings(gpu_ctx, scene_small):
    """OnValidate (Main.cs:206-209): an inspector value changed; the scene arrays are the same ones again."""
    cfg = h.small_config(88, 64)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    try:
        for _ in range(2):
            drv.render_frame(); orc.render_frame()
        cfg2 = h.small_config(88, 64)
        cfg2.TriReservoirTestsNum = 9
        cfg2.SpatialReuseIterations = 2      # even: no TransferToOriginal
        cfg2.DoBRDF = 0
        drv.reload(cfg2)
        orc.reload(None, cfg2)
        for f in range(2, 5):
            drv.render_frame(); orc.render_frame()
            compare(drv, orc, f"frame {f}")
    finally:
        drv.close()


def test_ris_project_reload(gpu_ctx):
    """The RIS project reloads on any camera or object move (BVH_RIS_DL Main.cs:117,122-130): FrameCount restarts, the
    buffers are made anew; its builder drops the old scene BVH first (MeshHelper.cs:371-372,418)."""
    scene = rt.build_scene_s0(subdivide=False)
    cfg = h.small_ris_config(72, 56)
    drv = rt.RisDriver(gpu_ctx, cfg, scene)
    orc = h.RisOracleRenderer(scene.arrays(), cfg)
    try:
        for _ in range(2):
            drv.render_frame(); orc.render_frame()
        n_bvs = len(scene.arrays()["BVs"])
        scene.set_object_transform(1, (-3.0, -2.0, 12.0), (0.0, 25.0, 0.0), (12.0, 12.0, 12.0))
        scene.rebuild(keep_old_scene_bvh=False)
        assert len(scene.arrays()["BVs"]) == n_bvs
        drv.reload()
        orc.reload(scene.arrays())
        for f in range(2, 5):
            if f == 2:   # the frame that reloads: the mesh BVHs below the replaced scene BVH are recognised, not walked again
                gpu_ctx.trace(True)
            drv.render_frame(); orc.render_frame()
            if f == 2:
                gpu_ctx.sync()
                gpu_ctx.trace(False)
                assert [l.split()[2] for l in gpu_ctx.trace_text().splitlines() if "scene_check" in l] == ["scene_check_reused"]
            for k in ("HitInfos", "Candidates"):
                assert np.array_equal(h.bits(drv.buffer(k)), h.bits(orc.buf[k])), f"frame {f}: {k}"
            for k in ("Result", "AccumulatedResult", "DebugOverlay", "RayHitPoints"):
                assert np.array_equal(h.bits(drv.texture(k)), h.bits(orc.tex[k])), f"frame {f}: {k}"
    finally:
        drv.close()


def test_released_handles_are_unbound(gpu_ctx):
    """ComputeBuffer.Release / RenderTexture.Release: a shader that still has the object bound must not touch it."""
    lib = gpu_ctx.lib
    pp = gpu_ctx.open_shader("PostProcessor")
    res = gpu_ctx.create_texture(32, 16, RT_FORMAT_RGBA32F)
    acc = gpu_ctx.create_texture(32, 16, RT_FORMAT_RGBA32F)
    try:
        gpu_ctx.check(lib.rt_set_texture(pp, 0, b"Result", res), "Result")
        gpu_ctx.check(lib.rt_set_texture(pp, 0, b"AccumulatedResult", acc), "AccumulatedResult")
        gpu_ctx.check(lib.rt_set_int(pp, b"FrameCount", 0), "FrameCount")
        gpu_ctx.check(lib.rt_dispatch(pp, 0, 4, 2, 1), "AccumulateFrames")
        lib.rt_texture_release(res)
        res = None
        assert lib.rt_dispatch(pp, 0, 4, 2, 1) == RT_ERR_UNBOUND
        assert b"Result" in lib.rt_last_error(gpu_ctx.h)
        pc = gpu_ctx.open_shader("PreCalc")
        tris = gpu_ctx.create_buffer(4, 28)
        verts = gpu_ctx.create_buffer(4, 20)
        gpu_ctx.check(lib.rt_set_buffer(pc, 0, b"Triangles", tris), "Triangles")
        gpu_ctx.check(lib.rt_set_buffer(pc, 0, b"Vertices", verts), "Vertices")
        gpu_ctx.check(lib.rt_set_int(pc, b"TrianglesNum", 4), "TrianglesNum")
        gpu_ctx.check(lib.rt_dispatch(pc, 0, 1, 1, 1), "CalcTriNormals")
        lib.rt_buffer_release(verts)
        assert lib.rt_dispatch(pc, 0, 1, 1, 1) == RT_ERR_UNBOUND
        lib.rt_buffer_release(tris)
        lib.rt_shader_close(pc)
    finally:
        if res:
            lib.rt_texture_release(res)
        lib.rt_texture_release(acc)
        lib.rt_shader_close(pp)


def test_flush_launches_what_is_held(gpu_ctx, scene_small):
    """A host that took the stream once and afterwards only dispatches and waits on that stream by its own means: the
    TraceRays of a frame without AccumulateFrames stays held until rt_flush."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    cfg = h.small_config(64, 48)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    try:
        drv.render_frame(); orc.render_frame()       # frame 0: buffers converted, tables built
        gpu_ctx.sync()
        stream = gpu_ctx.stream()
        (ptr, nbytes), = gpu_ctx.texture_rows(drv.texture_handle("Result"), 0, cfg.Resolution[1])
        before = orc.tex["Result"].copy()
        gpu_frame(drv, gpu_ctx, cfg, synced=False, post=False)
        orc_frame(orc, post=False)
        assert not np.array_equal(h.bits(before), h.bits(orc.tex["Result"]))
        got = np.empty_like(before)

        def raw_result():
            assert hip.hipStreamSynchronize(stream) == 0
            assert hip.hipMemcpy(got.ctypes.data, ptr, nbytes, 2) == 0   # hipMemcpyDeviceToHost
            return got

        # still held: the stream has nothing of TraceRays yet, Result is frame 0's
        assert np.array_equal(h.bits(raw_result()), h.bits(before))
        gpu_ctx.flush()
        assert np.array_equal(h.bits(raw_result()), h.bits(orc.tex["Result"]))
        # with merging switched off on the context nothing is ever held
        gpu_ctx.set_fusion(False)
        gpu_frame(drv, gpu_ctx, cfg, synced=False, post=False)
        orc_frame(orc, post=False)
        assert np.array_equal(h.bits(raw_result()), h.bits(orc.tex["Result"]))
    finally:
        gpu_ctx.set_fusion(True)
        gpu_ctx.set_tuning("frames_in_flight", 2)   # (this test no longer uses the stream it took: frames may alternate again)
        drv.close()


def _device_free_bytes():
    """hipMemGetInfo: free bytes of the current device."""
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


@pytest.mark.parametrize("pool_mb", [None, 0])
def test_reloads_reuse_released_device_memory(pool_mb):
    """A reload releases every buffer and creates buffers of the same sizes again: the library keeps what was released
    for them (device_pool_mb; 0 = every release is a hipFree), and the frames are the oracle's either way."""
    scene = rt.build_scene_s0(subdivide=False)
    cfg = h.small_config(512, 288)   # (per-pixel buffers of several MB: hipMemGetInfo does not see small blocks come and go)
    ctx = rt.RtContext(0)
    if pool_mb is not None:
        ctx.set_tuning("device_pool_mb", pool_mb)
    drv = rt.Driver(ctx, cfg, scene)
    orc = h.OracleRenderer(scene.arrays(), cfg)
    try:
        for _ in range(2):
            drv.render_frame(); orc.render_frame()
        free = []
        for k in range(3):
            drv.reload(); orc.reload(None)
            for _ in range(2):
                drv.render_frame(); orc.render_frame()
            compare(drv, orc, f"after reload {k}")
            free.append(_device_free_bytes())
        if pool_mb is None:
            # everything the second and third reload created came out of what the reload before had released
            assert abs(free[1] - free[0]) <= (2 << 20) and abs(free[2] - free[0]) <= (2 << 20), free
            held = free[0]
            ctx.set_tuning("device_pool_mb", 0)     # ... and what is kept idle goes back to the device on request
            assert _device_free_bytes() > held
            drv.reload(); orc.reload(None)
            drv.render_frame(); orc.render_frame()
            compare(drv, orc, "after a reload without the pool")
    finally:
        drv.close()
        ctx.close()


def test_device_pool_argument_is_checked(gpu_ctx):
    with pytest.raises(rt.NativeError):
        gpu_ctx.set_tuning("device_pool_mb", -1)


def _scene_checks(ctx, render):
    """Names of the scene checks (full: every mesh BVH walked; reused: only the scene BVH) logged while `render` runs."""
    ctx.trace(True)
    try:
        render()
        ctx.sync()
    finally:
        ctx.trace(False)
    return [l.split()[2] for l in ctx.trace_text().splitlines() if l.startswith("L ") and "scene_check" in l]


@pytest.mark.parametrize("reuse", [1, 0])
def test_reload_keeps_what_is_known_about_unchanged_meshes(reuse):
    """A reload uploads the mesh BVHs it had before and a new scene BVH above them: the library recognises the lower
    part of BVs by a digest taken on the device and walks only the scene BVH again (reuse_validation 0: every mesh)."""
    scene = rt.build_scene_s0(subdivide=False)
    cfg = h.small_config(96, 64)
    ctx = rt.RtContext(0)
    ctx.set_tuning("reuse_validation", reuse)
    drv = rt.Driver(ctx, cfg, scene)
    orc = h.OracleRenderer(scene.arrays(), cfg)
    try:
        assert _scene_checks(ctx, lambda: (drv.render_frame(), orc.render_frame())) == ["scene_check_full"]
        assert _scene_checks(ctx, lambda: (drv.render_frame(), orc.render_frame())) == []      # prepared already
        for k in range(3):
            scene.set_object_transform(1, (-3.0 + 0.2 * k, -2.0, 12.0), (0.0, 25.0 + 3 * k, 0.0), (12.0, 12.0, 12.0))
            scene.rebuild()
            drv.reload(); orc.reload(scene.arrays())
            got = _scene_checks(ctx, lambda: (drv.render_frame(), orc.render_frame()))
            assert got == (["scene_check_reused"] if reuse else ["scene_check_full"]), got
            drv.render_frame(); orc.render_frame()
            compare(drv, orc, f"after reload {k} (reuse_validation {reuse})")
    finally:
        drv.close()
        ctx.close()


def test_a_changed_scene_is_checked_in_full_again(scene_small):
    """What a reload may rely on is exactly: the same link words below the scene BVH, the scene BVH above them, meshes
    that were checked together, triangles still in the buffer.  Anything else is walked in full - and refused if bad."""
    a = scene_small.arrays()
    cfg = h.small_config(32, 32)
    ctx = rt.RtContext(0)
    drv = rt.Driver(ctx, cfg, scene_small)
    try:
        assert _scene_checks(ctx, drv.render_frame) == ["scene_check_full"]
        root = int(a["sceneBVHStartIndex"])
        mesh_root = int(a["SceneObjects"]["bvStartIndex"].max())
        assert a["BVs"]["indexA"][mesh_root] < 0 and mesh_root < root

        # the same arrays uploaded again: recognised
        ctx.set_buffer_data(drv.buffer_handle("BVs"), a["BVs"])
        assert _scene_checks(ctx, drv.render_frame) == ["scene_check_reused"]

        # a scene BVH that reaches down into a mesh: not the shape a reload has -> full walk, which refuses the shared nodes
        bad = a["BVs"].copy()
        bad["indexB"][root] = -(mesh_root + 1)
        ctx.set_buffer_data(drv.buffer_handle("BVs"), bad)
        with pytest.raises(rt.NativeError):
            drv.render_frame()
        ctx.set_buffer_data(drv.buffer_handle("BVs"), a["BVs"])
        assert _scene_checks(ctx, drv.render_frame) == ["scene_check_full"]     # (a failed preparation leaves nothing to rely on)

        # a mesh part that changed (here: damaged) is never taken for the known one
        bad = a["BVs"].copy()
        bad["indexB"][mesh_root] = bad["indexA"][mesh_root]
        ctx.set_buffer_data(drv.buffer_handle("BVs"), bad)
        with pytest.raises(rt.NativeError, match="reachable twice"):
            drv.render_frame()
        ctx.set_buffer_data(drv.buffer_handle("BVs"), a["BVs"])
        assert _scene_checks(ctx, drv.render_frame) == ["scene_check_full"]

        # an object that names a node inside a checked mesh as its root: that is not one of the meshes checked together
        objs = a["SceneObjects"].copy()
        child = -int(a["BVs"]["indexA"][mesh_root])
        assert a["BVs"]["indexA"][child] < 0
        k = int(np.argmax(objs["bvStartIndex"]))
        objs["bvStartIndex"][k] = child
        ctx.set_buffer_data(drv.buffer_handle("SceneObjects"), objs)
        assert _scene_checks(ctx, drv.render_frame) == ["scene_check_full"]     # (valid: half of the mesh as a mesh of its own)
        ctx.set_buffer_data(drv.buffer_handle("SceneObjects"), a["SceneObjects"])
        assert _scene_checks(ctx, drv.render_frame) == ["scene_check_full"]     # (the whole mesh was not among those last checked)
        ctx.set_buffer_data(drv.buffer_handle("BVs"), a["BVs"])
        assert _scene_checks(ctx, drv.render_frame) == ["scene_check_reused"]
        ctx.sync()
    finally:
        drv.close()
        ctx.close()
