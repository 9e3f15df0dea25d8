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


@pytest.mark.parametrize("unorm8,in_flight", [(0, 2), (1, 2), (0, 1), (0, 3), (1, 4)])
def test_overlapped_frames_equal_serial_frames_1080p(gpu_ctx, scene_80k, unorm8, in_flight):
    cfg = rt.default_config()
    cfg.Unorm8RenderTargets = unorm8
    frames = 7   # odd: the last frame leaves the buffers on their second copies
    try:
        gpu_ctx.set_tuning("frames_in_flight", in_flight)
        gpu_ctx.set_overlap(0)
        pos = np.array(cfg.cameraPosition[:], np.float32)
        rot = [float(x) for x in cfg.cameraRotation]

        def move(drv, f):   # a camera that drifts and turns: hit pixels come and go, TemporalReuse reprojects
            drv.set_camera(pos + np.float32(0.01 * f) * np.array([1, 0.2, 0.5], np.float32), h.quat((0.2, 1, 0), 0.9 * f, rot))

        serial = rt.Driver(gpu_ctx, cfg, scene_80k)
        for f in range(frames):
            move(serial, f)
            serial.render_frame()
        want = snapshot(serial)
        serial.close()
        gpu_ctx.set_overlap(1)
        beside = rt.Driver(gpu_ctx, cfg, scene_80k)
        for f in range(frames):
            move(beside, f)
            beside.render_frame()          # no call in between: frame f+1's kernels meet frame f's
        got = snapshot(beside)
        beside.close()
        for k in want:
            assert np.array_equal(h.bits(want[k]), h.bits(got[k])), k
        assert np.isfinite(got["DepthBuffer"]).sum() > 0.1 * got["DepthBuffer"].size
    finally:
        gpu_ctx.set_overlap(1)
        gpu_ctx.set_tuning("frames_in_flight", 2)


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


@pytest.mark.parametrize("in_flight,delay_us,mask", [(2, 0, 255), (3, 0, 255), (4, 0, 255), (1, 120, 255), (2, 60, 255), (2, 200, 255), (3, 150, 255),
                                                     (4, 90, 255), (2, 300, 1), (2, 300, 2), (2, 300, 128), (3, 300, 4), (3, 300, 3), (2, 300, 129)])
def test_unsynchronised_frames_against_the_oracle(gpu_ctx, scene_small, in_flight, delay_us, mask):
    """Frames enqueued without a single wait, then everything against the oracle; a moving camera makes the
    TemporalReuse of frame f read both hit-point targets, one of which frame f+1's primary rays overwrite.
    With delay_us the library holds its streams up for pseudo-random times before about every second launch (test aid
    "debug_stream_delay_us"), far longer than these small kernels run: the kernels of the frames in flight then meet in
    orders a quiet machine never produces, and only the waits the library enqueues keep the result right.  With a mask
    only some streams are held up (one dispatch stream, the primary rays' stream): the others then run ahead of them
    wherever nothing orders them."""
    cfg = h.small_config(200, 120)
    gpu_ctx.set_tuning("frames_in_flight", in_flight)
    gpu_ctx.set_tuning("debug_stream_delay_us", delay_us)
    gpu_ctx.set_tuning("debug_stream_delay_mask", mask)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    try:
        pos = np.array(cfg.cameraPosition[:], np.float32)
        rot = [float(x) for x in cfg.cameraRotation]
        for f in range(5):
            p = pos + np.float32(0.002 * f) * np.array([1, 0, 0.5], np.float32)
            drv.set_camera(p, rot); orc.set_camera(p, rot)
            drv.render_frame(); orc.render_frame()
        # then the camera turns away: pixels that held a hit keep the reservoirs of the last frame that hit them, which
        # by then live in one or the other of the two copies the frames in flight alternate between
        for f in range(3):
            r = h.quat((0.3, 1, 0), 35.0 * (f + 1), rot)
            drv.set_camera(pos, r); orc.set_camera(pos, r)
            drv.render_frame(); orc.render_frame()
        # (some pixels that held a hit when the camera looked at the scene now see the sky, and some still hit)
        sky = ~np.isfinite(orc.tex["DepthBuffer"])
        assert sky.sum() > 0.2 * sky.size and (orc.buf["Candidates"]["totCandidates"].reshape(sky.shape)[sky] > 0).any()
        for k in BUFS:
            assert np.array_equal(h.bits(drv.buffer(k)), h.bits(orc.buf[k])), k
        for k in TEXS:
            assert np.array_equal(h.bits(drv.texture(k)), h.bits(orc.tex[k])), k
    finally:
        gpu_ctx.set_tuning("frames_in_flight", 2)
        gpu_ctx.set_tuning("debug_stream_delay_us", 0)
        drv.close()


@pytest.mark.parametrize("delay_us,in_flight", [(0, 3), (100, 3), (150, 2), (0, 1), (120, 4)])
def test_ris_frames_overlap_and_match(gpu_ctx, scene_small, delay_us, in_flight):
    cfg = h.small_ris_config(120, 88)
    gpu_ctx.set_tuning("debug_stream_delay_us", delay_us)
    gpu_ctx.set_tuning("ris_frames_in_flight", in_flight)
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
        for _ in range(7):
            drv.render_frame(); orc.render_frame()
        for k in ("HitInfos", "Candidates"):
            assert np.array_equal(h.bits(drv.buffer(k)), h.bits(orc.buf[k])), k
        for k in ("Result", "AccumulatedResult", "DebugOverlay", "RayHitPoints"):
            assert np.array_equal(h.bits(drv.texture(k)), h.bits(orc.tex[k])), k
    finally:
        gpu_ctx.set_tuning("debug_stream_delay_us", 0)
        gpu_ctx.set_tuning("ris_frames_in_flight", 3)
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
                                        ("overlap_behind_candidates", 0), ("region_blocks_rays", 1), ("region_blocks_rays", 4),
                                        ("region_blocks_reuse", 2), ("region_blocks_primary", 8), ("one_wave_groups", 0), ("one_wave_groups", 5), ("one_wave_groups", 31)])
def test_speed_only_choices_leave_every_bit_alone(gpu_ctx, scene_small, name, value):
    """rt_set_tuning: the alternatives that were measured and not adopted still render the same frames - the LDS-staged
    SpatialReusePass (also with a window of +-1, and with blocks at x < MaxOffset whose neighbours wrap into the row
    above), primary rays as a background launch, primary rays started as early as their inputs allow, other region
    sizes for the three heaviest-first workgroup tables (the dispatches merged into one launch then ask for different
    tables of the same grid), per-pixel kernels with four waves or one wave per workgroup."""
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
        with pytest.raises(rt.NativeError):
            gpu_ctx.set_tuning("region_blocks_rays", 3)
    finally:
        for n, v in (("spatial_lds_max_offset", 0), ("primary_background_groups", 0), ("overlap_behind_candidates", 1),
                     ("region_blocks_rays", 2), ("region_blocks_reuse", 4), ("region_blocks_primary", 4), ("one_wave_groups", 15)):
            gpu_ctx.set_tuning(n, v)


@pytest.mark.parametrize("delay_us", [0, 120])
def test_two_drivers_taking_turns_on_one_context(gpu_ctx, scene_small, delay_us):
    """Frames of two frame loops enqueued alternately, and two in a row: each loop's frames equal the frames it renders
    on its own (the copies and streams the frames alternate between belong to the context, the buffers to the loops)."""
    cfg_a, cfg_b = h.small_config(136, 96), h.small_config(96, 136)
    order = "ababbaabab"
    want = {}
    for name, cfg in (("a", cfg_a), ("b", cfg_b)):
        alone = rt.Driver(gpu_ctx, cfg, scene_small)
        for _ in range(order.count(name)):
            alone.render_frame()
        want[name] = snapshot(alone)
        alone.close()
    drv = {"a": rt.Driver(gpu_ctx, cfg_a, scene_small), "b": rt.Driver(gpu_ctx, cfg_b, scene_small)}
    try:
        gpu_ctx.set_tuning("debug_stream_delay_us", delay_us)
        for name in order:
            drv[name].render_frame()
        for name in "ab":
            got = snapshot(drv[name])
            for k in got:
                assert np.array_equal(h.bits(want[name][k]), h.bits(got[k])), f"{name}: {k}"
    finally:
        gpu_ctx.set_tuning("debug_stream_delay_us", 0)
        for d in drv.values():
            d.close()


@pytest.mark.parametrize("in_flight", [2, 3])
def test_uploads_between_frames_in_flight(gpu_ctx, scene_small, in_flight):
    """A host that reads a per-pixel object back and uploads it again between frames changes nothing: the reservoirs come
    back merged from their two copies, the upload makes one copy of them, and the frames go on alternating."""
    cfg = h.small_config(152, 104)
    gpu_ctx.set_tuning("frames_in_flight", in_flight)
    plain, busy = rt.Driver(gpu_ctx, cfg, scene_small), rt.Driver(gpu_ctx, cfg, scene_small)
    try:
        pos = np.array(cfg.cameraPosition[:], np.float32)
        rot = [float(x) for x in cfg.cameraRotation]
        for f in range(7):
            r = h.quat((0.3, 1, 0), 14.0 * f, rot)
            for d in (plain, busy):
                d.set_camera(pos, r)
            plain.render_frame()
            busy.render_frame()
            if f == 2:
                gpu_ctx.set_buffer_data(busy.buffer_handle("Candidates"), busy.buffer("Candidates"))
            if f == 3:
                gpu_ctx.set_buffer_data(busy.buffer_handle("HitInfos"), busy.buffer("HitInfos"))
            if f == 4:
                depth = np.ascontiguousarray(busy.texture("DepthBuffer"))
                gpu_ctx.check(gpu_ctx.lib.rt_texture_set_data(busy.texture_handle("DepthBuffer"), depth.ctypes.data, depth.nbytes), "rt_texture_set_data")
            if f == 5:
                gpu_ctx.buffer_rows(busy.buffer_handle("CandidatesB"), cfg.Resolution[0], 8, 40)
        want, got = snapshot(plain), snapshot(busy)
        for k in want:
            assert np.array_equal(h.bits(want[k]), h.bits(got[k])), k
    finally:
        gpu_ctx.set_tuning("frames_in_flight", 2)
        plain.close()
        busy.close()


def test_weight_recalc_orders_the_frames_in_flight(gpu_ctx, scene_small):
    """WEIGHT_RECALC: TraceRays writes into TemporalFrameBuffer (ReSTIR_BVH_RayTracer.compute:645-657), which the next
    frame's TemporalReuse reads: the next frame's reuse then has to wait for it."""
    cfg = h.small_config(168, 96)
    cfg.DoWeightRecalc = 1
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    orc = h.OracleRenderer(scene_small.arrays(), cfg)
    try:
        for _ in range(5):
            drv.render_frame(); orc.render_frame()
        for k in BUFS:
            assert np.array_equal(h.bits(drv.buffer(k)), h.bits(orc.buf[k])), k
        for k in TEXS:
            assert np.array_equal(h.bits(drv.texture(k)), h.bits(orc.tex[k])), k
    finally:
        drv.close()


@pytest.mark.parametrize("in_flight", [2, 3, 4])
def test_ris_candidates_read_back_from_the_copy_that_holds_them(gpu_ctx, scene_small, in_flight):
    """The RIS project's Candidates get a copy per frame in flight, tagged per pixel ray.  Frames that render different
    row bands leave every pixel's newest candidates in whichever copy the last frame that covered it wrote: the
    readback has to pick per pixel, and equals what one buffer holds when the same frames run one after the other."""
    cfg = h.small_ris_config(96, 80)
    bands = [(0, 48), (32, 80), (16, 64), (0, 80), (40, 80), (0, 24), (8, 72)]
    got = {}
    try:
        for mode in ("serial", "in flight"):
            gpu_ctx.set_overlap(0 if mode == "serial" else 2)
            gpu_ctx.set_tuning("ris_frames_in_flight", 1 if mode == "serial" else in_flight)
            drv = rt.RisDriver(gpu_ctx, cfg, scene_small)
            for b in bands:
                gpu_ctx.set_row_band(*b)
                drv.render_frame()
            gpu_ctx.set_row_band(0, 0)
            got[mode] = drv.buffer("Candidates")
            if mode == "in flight":   # an upload makes one copy of them again, and the frames go on
                gpu_ctx.set_buffer_data(drv.buffer_handle("Candidates"), got[mode])
                drv.render_frame(); drv.render_frame()
                gpu_ctx.sync()
            drv.close()
        assert np.array_equal(h.bits(got["serial"]), h.bits(got["in flight"]))
        assert np.any(h.bits(got["serial"]))
    finally:
        gpu_ctx.set_row_band(0, 0)
        gpu_ctx.set_overlap(1)
        gpu_ctx.set_tuning("ris_frames_in_flight", 3)
