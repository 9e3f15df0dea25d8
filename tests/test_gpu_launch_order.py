"""Are the launches of frames in flight ordered wherever they have to be?  Decided from the library's own log of what
it put on which stream (rt_debug_trace): every launch with the per-pixel objects it reads and writes, every event
record and wait, every host operation.  harness.launch_order_violations rebuilds the happens-before relation from
stream order and record -> wait edges and reports every pair of launches that touch the same object, one of them
writing, and are not ordered by it.  Timing plays no part: what this finds (or does not find) does not depend on how
the streams happened to be scheduled on the test machine."""
import numpy as np
import pytest

import harness as h

rt = h.rt
pytestmark = pytest.mark.gpu


def traced(ctx, body):
    ctx.sync()
    ctx.trace(True)
    try:
        body()
        ctx.sync()
    finally:
        ctx.trace(False)
    text = ctx.trace_text()
    assert text.count("\n") > 10
    return text


@pytest.mark.parametrize("in_flight,behind,fuse,weight_recalc,refresh",
                         [(2, 1, 1, 0, 8), (3, 1, 1, 0, 8), (4, 1, 1, 0, 8), (2, 0, 1, 0, 8), (3, 0, 1, 1, 8), (2, 1, 0, 0, 8), (1, 0, 1, 0, 8),
                          (4, 1, 1, 0, 4), (4, 0, 1, 0, 5), (3, 1, 1, 0, 3), (2, 1, 1, 0, 1), (4, 1, 1, 0, 2), (4, 0, 1, 0, 1), (3, 1, 1, 0, 1)])
def test_frames_in_flight_are_ordered(gpu_ctx, scene_small, in_flight, behind, fuse, weight_recalc, refresh):
    """`refresh`: the heaviest-first workgroup tables are ranked anew every so many frames, into the pair of tables that
    is not in use - with short periods while frames that walk the old pair are still in flight (the library must then
    notice that the spare pair may still be read, and wait)."""
    cfg = h.small_config(136, 104)
    cfg.DoWeightRecalc = weight_recalc
    try:
        gpu_ctx.set_tuning("table_refresh_frames", refresh)
        gpu_ctx.set_tuning("frames_in_flight", in_flight)
        gpu_ctx.set_tuning("overlap_behind_candidates", behind)
        gpu_ctx.set_fusion(bool(fuse))
        drv = rt.Driver(gpu_ctx, cfg, scene_small)
        for _ in range(3):
            drv.render_frame()

        def body():
            for f in range(19):                # (the region order is rebuilt every eighth frame: twice in here)
                drv.render_frame()
                if f == 9:
                    drv.buffer("Candidates")   # a readback between frames
                if f == 13:
                    gpu_ctx.buffer_rows(drv.buffer_handle("CandidatesB"), cfg.Resolution[0], 8, 24)

        text = traced(gpu_ctx, body)
        streams = {int(line.split()[1]) for line in text.splitlines() if line.startswith("L")}
        if fuse:
            assert streams >= set(range(in_flight)) | {4}, streams      # the frames did alternate, the primary rays had their stream
        bad = h.launch_order_violations(text)
        assert not bad, "\n".join(bad)
        drv.close()
    finally:
        gpu_ctx.set_tuning("table_refresh_frames", 8)
        gpu_ctx.set_tuning("frames_in_flight", 2)
        gpu_ctx.set_tuning("overlap_behind_candidates", 1)
        gpu_ctx.set_fusion(True)


def test_two_frame_loops_and_the_ris_project_are_ordered(gpu_ctx, scene_small):
    a, b = rt.Driver(gpu_ctx, h.small_config(96, 72), scene_small), rt.Driver(gpu_ctx, h.small_config(72, 96), scene_small)
    ris = rt.RisDriver(gpu_ctx, h.small_ris_config(80, 64), scene_small)
    try:
        for d in (a, b, ris):
            d.render_frame()

        def body():
            for name in "abbarabrrabaab":
                {"a": a, "b": b, "r": ris}[name].render_frame()

        bad = h.launch_order_violations(traced(gpu_ctx, body))
        assert not bad, "\n".join(bad)
    finally:
        for d in (a, b, ris):
            d.close()


def test_banded_frames_are_ordered(gpu_ctx, scene_small):
    """rt_set_overlap 2: InitialTrace on a row band changes copies too (what a rank of the multi-GPU loop does)."""
    cfg = h.small_config(128, 160)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    try:
        gpu_ctx.set_overlap(2)
        gpu_ctx.set_row_band(48, 112)
        for _ in range(2):
            drv.render_frame()

        def body():
            for _ in range(10):
                drv.update_per_frame()
                drv.run_stage(0)
                drv.run_stage(1)
                gpu_ctx.set_row_band(56, 104); drv.run_spatial_pass(0)
                gpu_ctx.set_row_band(48, 56); drv.run_spatial_pass(0)
                gpu_ctx.set_row_band(104, 112); drv.run_spatial_pass(0)
                gpu_ctx.set_row_band(48, 112)
                drv.run_spatial_pass(1); drv.run_spatial_pass(2)
                drv.run_transfer()
                drv.run_stage(3)
                drv.run_post_processing()

        bad = h.launch_order_violations(traced(gpu_ctx, body))
        assert not bad, "\n".join(bad)
    finally:
        gpu_ctx.set_row_band(0, 0)
        gpu_ctx.set_overlap(1)
        drv.close()


def test_a_reload_between_frames_in_flight_is_ordered(gpu_ctx):
    """Main.Update -> SetData while frames are in flight: the releases, uploads and the rebuilt traversal arrays are host
    operations; everything launched before them must be behind them and everything after them behind them too."""
    scene = rt.build_scene_s0(subdivide=False)
    cfg = h.small_config(104, 72)
    drv = rt.Driver(gpu_ctx, cfg, scene)
    try:
        for _ in range(3):
            drv.render_frame()

        def body():
            for f in range(10):
                if f == 4:
                    scene.set_object_transform(1, (-3.0, -2.0, 12.0), (0.0, 25.0, 0.0), (12.0, 12.0, 12.0))
                    scene.rebuild()
                    drv.reload()
                drv.render_frame()

        text = traced(gpu_ctx, body)
        assert text.count("\nH 0") > 10          # the reload's copies and conversions
        bad = h.launch_order_violations(text)
        assert not bad, "\n".join(bad)
    finally:
        drv.close()
