"""rt_dispatch_no_alloc (the dispatch that allocates no device memory) against rt_dispatch, through the C-ABI.

Two drivers render the same frames on one context: one with rt_dispatch (the C++ frame driver), the other with the
frame's dispatches issued from here through rt_dispatch_no_alloc, in the order Main.RunReSTIRShader makes them
(Main.cs:387-420).  Every per-pixel buffer and render target must agree bit for bit.  Then the same again at a smaller
resolution: all calls on a GPU share one scratch array, which still holds the larger launch's table and hit counts.
"""
import numpy as np
import pytest

import harness as h

rt = h.rt
pytestmark = pytest.mark.gpu

BUFFERS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXTURES = ("DepthBuffer", "NormalsBuffer", "DebugOverlay", "RayHitPointsA", "RayHitPointsB", "Result", "AccumulatedResult")
RT_ERR_LIMIT, RT_ERR_NOT_PREPARED, RT_ERR_BUSY = -6, -7, -9


def device_free_bytes():
    """hipMemGetInfo: the free bytes of the current device (an allocation of any size moves it)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def no_alloc_frame(ctx, drv, cfg):
    """One frame of Main.RunReSTIRShader + RunPostProcessingShader with every dispatch made by rt_dispatch_no_alloc."""
    lib = ctx.lib
    W, H = cfg.Resolution[0], cfg.Resolution[1]
    gx, gy = (W + 7) // 8, (H + 7) // 8
    sh, pp = drv.shader("rt"), drv.shader("pp")

    def dispatch(shader, kernel):
        ctx.check(lib.rt_dispatch_no_alloc(shader, kernel, gx, gy, 1), f"rt_dispatch_no_alloc({kernel})")

    drv.update_per_frame()
    dispatch(sh, 0)                                       # InitialTrace
    if cfg.TemporalReuseWeight > 0:
        dispatch(sh, 3)                                   # TemporalReuse
    cycle = False
    for i in range(cfg.SpatialReuseIterations):           # Main.SpatialReuse, :387-402
        cycle = not cycle
        ctx.check(lib.rt_set_int(sh, b"MaxOffset", int(np.float32(3.0 ** i)) * cfg.SpatialReuseBlur), "MaxOffset")
        ctx.check(lib.rt_set_bool(sh, b"ReuseBufferCycle", 1 if cycle else 0), "ReuseBufferCycle")
        dispatch(sh, 1)                                   # SpatialReusePass
    if cycle:
        dispatch(sh, 2)                                   # TransferToOriginal
    dispatch(sh, 4)                                       # TraceRays
    dispatch(pp, 0)                                       # AccumulateFrames


def snapshot(drv):
    drv.ctx.sync()
    return {**{n: drv.buffer(n) for n in BUFFERS}, **{n: drv.texture(n) for n in TEXTURES}}


def run_pair(ctx, scene, res, frames):
    cfg = h.small_config(*res)
    ref = rt.Driver(ctx, cfg, scene)
    want = []
    try:
        for frame in range(frames + 1):
            ref.render_frame()
            want.append(snapshot(ref))
    finally:
        ref.close()
    sut = rt.Driver(ctx, cfg, scene)
    try:
        # frame 0 through rt_dispatch: builds the traversal arrays and converts the per-pixel buffers
        sut.render_frame()
        ctx.sync()
        for frame in range(1, frames + 1):
            free_before = device_free_bytes()
            no_alloc_frame(ctx, sut, cfg)
            ctx.sync()
            assert device_free_bytes() == free_before, "a no-alloc frame changed the device's free memory"
            got = snapshot(sut)
            for k in got:
                assert np.array_equal(h.bits(want[frame][k]), h.bits(got[k])), f"{res} frame {frame}: {k} differs"
        assert np.isfinite(got["DepthBuffer"]).sum() > 0.1 * got["DepthBuffer"].size
    finally:
        sut.close()


def test_no_alloc_dispatch_matches_rt_dispatch_twice_second_smaller(gpu_ctx, scene_small):
    run_pair(gpu_ctx, scene_small, (200, 120), frames=2)   # 13 x 8 blocks, ragged edges
    run_pair(gpu_ctx, scene_small, (64, 48), frames=2)     # smaller: the scratch still holds the larger launch's data


def test_no_alloc_refuses_what_needs_an_allocation(gpu_ctx, scene_small):
    lib = gpu_ctx.lib
    cfg = h.small_config(64, 64)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    try:
        drv.update_per_frame()
        # nothing dispatched yet: the traversal arrays do not exist and the buffers are in the ABI layout
        assert lib.rt_dispatch_no_alloc(drv.shader("rt"), 0, 8, 8, 1) == RT_ERR_NOT_PREPARED
        assert b"allocates" in lib.rt_last_error(gpu_ctx.h)
        assert lib.rt_dispatch_no_alloc(drv.shader("rt"), 1, 8, 8, 1) == RT_ERR_NOT_PREPARED
        drv.run_stage(0)
        drv.run_stage(1)
        drv.run_stage(2)
        drv.run_stage(3)
        drv.run_post_processing()
        no_alloc_frame(gpu_ctx, drv, cfg)                 # now everything is in place
        # an upload puts the buffer back into the ABI layout
        gpu_ctx.set_buffer_data(drv.buffer_handle("Candidates"), drv.buffer("Candidates"))
        assert lib.rt_dispatch_no_alloc(drv.shader("rt"), 3, 8, 8, 1) == RT_ERR_NOT_PREPARED
        # beyond the scratch table: 4096 regions of 64x64 pixels
        sh = gpu_ctx.open_shader("ReSTIR_BVH_RayTracer")
        import ctypes as C
        gpu_ctx.check(lib.rt_set_ints(sh, b"Resolution", (C.c_int * 2)(8192, 4160), 2), "Resolution")
        assert lib.rt_dispatch_no_alloc(sh, 1, 1024, 520, 1) == RT_ERR_LIMIT
        assert b"4096" in lib.rt_last_error(gpu_ctx.h)
        lib.rt_shader_close(sh)
    finally:
        drv.close()


def test_two_scenes_on_one_context_stay_prepared(gpu_ctx, scene_small):
    """The context keeps the traversal arrays of several sets of scene buffers: two drivers (each with its own upload of
    the scene) alternate frames, and neither's rt_dispatch_no_alloc finds its arrays gone."""
    cfg = h.small_config(64, 48)
    a, b = rt.Driver(gpu_ctx, cfg, scene_small), rt.Driver(gpu_ctx, cfg, scene_small)
    try:
        a.render_frame()
        b.render_frame()
        gpu_ctx.sync()
        for _ in range(2):
            no_alloc_frame(gpu_ctx, a, cfg)
            no_alloc_frame(gpu_ctx, b, cfg)
        sa, sb = snapshot(a), snapshot(b)
        for k in sa:
            assert np.array_equal(h.bits(sa[k]), h.bits(sb[k])), k
    finally:
        a.close()
        b.close()


def test_second_context_is_refused_while_the_scratch_is_in_use(gpu_ctx, scene_80k):
    """Every context of a GPU shares the scratch array: a no-alloc dispatch from another context is refused while the
    first one's launches are still in flight, and accepted once they are done."""
    lib = gpu_ctx.lib
    cfg = h.small_config(1920, 1080)
    other = rt.RtContext(0)
    drv = rt.Driver(gpu_ctx, cfg, scene_80k)
    try:
        drv.render_frame()
        gpu_ctx.sync()
        for _ in range(6):                       # several milliseconds of work on gpu_ctx's stream
            no_alloc_frame(gpu_ctx, drv, cfg)
        sh = other.open_shader("PostProcessor")
        rc = lib.rt_dispatch_no_alloc(sh, 0, 8, 8, 1)
        assert rc == RT_ERR_BUSY, rc
        assert b"scratch" in lib.rt_last_error(other.h)
        gpu_ctx.sync()
        assert lib.rt_dispatch_no_alloc(sh, 0, 8, 8, 1) != RT_ERR_BUSY   # (unbound textures: another error, not this one)
        lib.rt_shader_close(sh)
    finally:
        drv.close()
        other.close()
