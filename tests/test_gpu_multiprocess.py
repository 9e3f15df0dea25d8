"""Rehearsal of the multi-process GPU path on ONE GPU: two ranks share cuda:0, each renders its row band through
the C-ABI, halos and the frame buffer travel over gloo (staged through host memory, because RCCL refuses two ranks
on one device).  Rank 0 checks the gathered frame against a single-process full-frame render, bit for bit."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, res, frames, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import harness as h
    rt = h.rt
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    scene = rt.build_scene_s0(subdivide=False, threads=2)
    cfg = h.small_config(*res)
    runner = mg.DistributedRunner(rt, scene, cfg, rank, world, 0, backend="gloo")
    try:
        ref_ctx = ref = None
        if rank == 0:
            ref_ctx = rt.RtContext(0)
            ref = rt.Driver(ref_ctx, cfg, scene)
        for frame in range(frames):
            runner.render_frame()
            runner.sync()
            if rank == 0:
                ref.render_frame()
                ref_ctx.sync()
                assert np.array_equal(h.bits(runner.drv.texture("Result")), h.bits(ref.texture("Result"))), frame
                b1 = runner.bounds[1]
                assert np.array_equal(h.bits(runner.drv.texture("AccumulatedResult")[:b1]),
                                      h.bits(ref.texture("AccumulatedResult")[:b1])), frame
                lo, hi = 0, b1 * res[0]
                for name in ("Candidates", "TemporalFrameBuffer", "HitInfos"):
                    assert np.array_equal(h.bits(runner.drv.buffer(name)[lo:hi]), h.bits(ref.buffer(name)[lo:hi])), (frame, name)
        runner.barrier()
        if rank == 0:
            ref.close()
            ref_ctx.close()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        runner.close()


def test_two_ranks_one_gpu(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, (256, 192), 3, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_native_frame_loop_single_rank(gpu_ctx, scene_small):
    """The C++ multi-GPU frame loop with one rank (and a one-rank RCCL communicator, which exercises the RCCL set-up
    and tear-down on this GPU): must equal the plain driver bit for bit."""
    import harness as h
    rt = h.rt
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    cfg = h.small_config(128, 96)
    ref = rt.Driver(gpu_ctx, cfg, scene_small)
    for _ in range(3):
        ref.render_frame()
    want = {n: ref.texture(n) for n in ("Result", "AccumulatedResult")}
    want_c = ref.buffer("Candidates")
    ref.close()
    runner = mg.RcclRunner(rt, scene_small, cfg, 0, 1, 0, force_comm=True)
    try:
        assert runner.bounds == [0, 96]
        runner.render_frames(1)
        t = runner.timed_frames(1)
        # (a one-rank communicator takes the loop's multi-rank path with nothing to transfer: the two stamps around the
        # empty exchange are microseconds apart)
        assert t["frame"] > 0 and t["halo_exchange"] < 0.05 and t["gather"] < 0.05 and t["before_halo"] + t["interior_rows"] + t["after_halo"] <= t["frame"] * 1.05
        runner.render_frame()
        runner.sync()
        for n in want:
            assert np.array_equal(h.bits(runner.drv.texture(n)), h.bits(want[n])), n
        assert np.array_equal(h.bits(runner.drv.buffer("Candidates")), h.bits(want_c))
        # the halo transfer path itself, looped back to this rank: rows 40..52 -> rows 8..20
        depth = runner.drv.texture("DepthBuffer")
        assert runner.host.rth_mg_loopback(runner.h, 40, 52, 8) == 0, runner.host.rth_mg_last_error(runner.h)
        runner.sync()
        got_c, got_d = runner.drv.buffer("Candidates"), runner.drv.texture("DepthBuffer")
        W = 128
        assert np.array_equal(h.bits(got_c[8 * W:20 * W]), h.bits(want_c[40 * W:52 * W]))
        assert np.array_equal(h.bits(got_c[20 * W:]), h.bits(want_c[20 * W:])) and np.array_equal(h.bits(got_c[:8 * W]), h.bits(want_c[:8 * W]))
        assert np.array_equal(h.bits(got_d[8:20]), h.bits(depth[40:52])) and np.array_equal(h.bits(got_d[20:]), h.bits(depth[20:]))
        assert np.isfinite(depth[40:52]).any()
    finally:
        runner.close()


@pytest.mark.parametrize("world,res,balanced", [(2, (128, 96), False), (3, (96, 160), True), (8, (64, 384), False), (2, (64, 96), False)])
def test_native_schedule_many_ranks_in_process(scene_small, world, res, balanced):
    """The C++ banded schedule for 2, 3 and 8 ranks, all held in this process on the one GPU: each rank is stepped
    through the four steps of a frame (rth_mg_phase) and the transfer lists the frame loop would hand to RCCL
    (rth_mg_plan) are carried out here with device-to-device copies, the k-th send of rank a to rank b matched with
    the k-th receive of rank b from rank a, as RCCL matches them.  The copies are made as LATE as the native loop's
    stream ordering allows: the halos after the first spatial pass has run on the interior rows, the gather of a frame
    after the next frame's spatial reuse, just before its TraceRays.  Rank 0's gathered image and every rank's band
    must equal the single-GPU render bit for bit over several frames.  (The last case has bands of 48 rows, barely
    more than the 42 halo rows: the spatial passes of each rank recompute most of its neighbour's band.)"""
    import ctypes as C
    import harness as h
    rt = h.rt
    bindings = importlib.import_module("ray-tracer_amd.bindings")
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    W, H = res
    cfg = h.small_config(W, H)
    host = rt.host_lib()
    ctxs = [rt.RtContext(0) for _ in range(world + 1)]
    drvs = [rt.Driver(c, cfg, scene_small) for c in ctxs]
    ref_ctx, ref = ctxs[world], drvs[world]
    ranks = []
    err = C.create_string_buffer(512)
    try:
        for r in range(world):
            g = host.rth_mg_create(drvs[r].h, r, world, None, None, err, 512)
            assert g, err.value.decode()
            ranks.append(g)
        _, _, total = mg.native_halo_plan(cfg.SpatialReuseIterations, cfg.SpatialReuseBlur)
        if balanced:
            # what rth_mg_balance does, with the all-reduce made here: each rank measures its own rows, the vectors
            # are summed, every rank derives the same cuts from the sum
            summed = np.zeros(H)
            for g in ranks:
                mine = (C.c_double * H)()
                assert host.rth_mg_row_costs(g, mine) == 0, host.rth_mg_last_error(g)
                summed += np.array(mine)
            assert (summed >= W).all() and summed.max() > W      # every row measured by exactly one rank, some rows hit
            cuts = mg.native_balanced_bounds(H, world, summed, total)
            assert cuts == mg.band_bounds(H, world, summed, total) and cuts != mg.band_bounds(H, world)
            for g in ranks:
                assert host.rth_mg_set_bounds(g, (C.c_int * (world + 1))(*cuts)) == 0, host.rth_mg_last_error(g)
            assert host.rth_mg_set_bounds(ranks[0], (C.c_int * (world + 1))(*([0, 8, 16, H][:world + 1]))) != 0   # shorter than the halo
        b = (C.c_int * (world + 1))()
        host.rth_mg_bounds(ranks[0], b)
        cuts = list(b)
        # a rank without a communicator cannot run the RCCL loop
        assert host.rth_mg_render_frames(ranks[0], 1) != 0

        def plan(which):
            plans = []
            for g in ranks:
                buf = (bindings.Xfer * 64)()
                n = host.rth_mg_plan(g, which, buf, 64)
                assert n >= 0, host.rth_mg_last_error(g)
                plans.append([buf[i] for i in range(n)])
            return plans

        def carry_out(plans):
            for c in ctxs[:world]:
                c.sync()
            moved = 0
            for a in range(world):
                for bb in range(world):
                    sends = [x for x in plans[a] if x.send and x.peer == bb]
                    recvs = [x for x in plans[bb] if not x.send and x.peer == a]
                    assert len(sends) == len(recvs), (a, bb)
                    for s, d in zip(sends, recvs):
                        assert s.bytes == d.bytes and s.bytes > 0
                        assert hip.hipMemcpy(d.ptr, s.ptr, s.bytes, 3) == 0
                        moved += s.bytes
            return moved

        def phase(k):
            for g in ranks:
                assert host.rth_mg_phase(g, k) == 0, host.rth_mg_last_error(g)

        gather, want_image = None, None
        for frame in range(3):
            phase(0)
            halo = plan(0)
            phase(1)                            # the interior rows of the first pass: before any halo has arrived
            halo_bytes = carry_out(halo)
            phase(2)
            if gather is not None:              # the previous frame's image arrives only now
                assert carry_out(gather) == (H - cuts[1]) * W * 16
                assert np.array_equal(h.bits(drvs[0].texture("Result")[cuts[1]:]), h.bits(want_image[cuts[1]:])), frame - 1
            phase(3)
            gather = plan(1)
            for c in ctxs[:world]:
                c.sync()
            ref.render_frame()
            ref_ctx.sync()
            want_image = ref.texture("Result")
            for r in range(world):
                lo, hi = cuts[r], cuts[r + 1]
                for name in ("Result", "AccumulatedResult", "DepthBuffer"):
                    assert np.array_equal(h.bits(drvs[r].texture(name)[lo:hi]), h.bits(ref.texture(name)[lo:hi])), (frame, r, name)
                for name in ("Candidates", "CandidatesB", "TemporalFrameBuffer", "HitInfos"):
                    assert np.array_equal(h.bits(drvs[r].buffer(name)[lo * W:hi * W]), h.bits(ref.buffer(name)[lo * W:hi * W])), (frame, r, name)
            # volume: every interior cut carries `total` rows each way of the reservoir records and a 4-byte depth
            per_pixel, rest = divmod(halo_bytes, 2 * (world - 1) * total * W)
            assert rest == 0 and 52 + 4 <= per_pixel <= 64 + 4, halo_bytes
        carry_out(gather)
        assert np.array_equal(h.bits(drvs[0].texture("Result")), h.bits(want_image))
    finally:
        for g in ranks:
            host.rth_mg_destroy(g)
        for d in drvs:
            d.close()
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("world,res", [(2, (64, 48)), (5, (96, 120))])
def test_native_ris_schedule_many_ranks_in_process(scene_small, world, res):
    """The banded RIS frame loop (rth_mg_create_ris) for several ranks held in this process: phase 0 generates the
    band's candidates, phase 3 traces and accumulates, plan 0 (the halo) is empty, plan 1 is the gather of Result,
    carried out here with device copies."""
    import ctypes as C
    import harness as h
    rt = h.rt
    bindings = importlib.import_module("ray-tracer_amd.bindings")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    W, H = res
    cfg = h.small_ris_config(W, H)
    tot = cfg.SceneObjectCandidatesNum * cfg.TriCandidatesNum
    host = rt.host_lib()
    ctxs = [rt.RtContext(0) for _ in range(world + 1)]
    drvs = [rt.RisDriver(c, cfg, scene_small) for c in ctxs]
    ref_ctx, ref = ctxs[world], drvs[world]
    ranks = []
    err = C.create_string_buffer(512)
    try:
        for r in range(world):
            g = host.rth_mg_create_ris(drvs[r].h, r, world, None, None, err, 512)
            assert g, err.value.decode()
            ranks.append(g)
        b = (C.c_int * (world + 1))()
        host.rth_mg_bounds(ranks[0], b)
        cuts = list(b)
        for frame in range(2):
            plans = []
            for g in ranks:
                assert host.rth_mg_phase(g, 0) == 0, host.rth_mg_last_error(g)
                assert host.rth_mg_plan(g, 0, None, 0) == 0
                assert host.rth_mg_phase(g, 1) == 0 and host.rth_mg_phase(g, 2) == 0     # nothing to do: no spatial reuse
                assert host.rth_mg_phase(g, 3) == 0, host.rth_mg_last_error(g)
                buf = (bindings.Xfer * 16)()
                n = host.rth_mg_plan(g, 1, buf, 16)
                assert n == (world - 1 if g is ranks[0] else 1), host.rth_mg_last_error(g)
                plans.append([buf[i] for i in range(n)])
            for c in ctxs[:world]:
                c.sync()
            for r in range(1, world):
                (s,) = plans[r]
                d = [x for x in plans[0] if x.peer == r][0]
                assert s.send and s.peer == 0 and not d.send and s.bytes == d.bytes == (cuts[r + 1] - cuts[r]) * W * 16
                assert hip.hipMemcpy(d.ptr, s.ptr, s.bytes, 3) == 0
            ref.render_frame()
            ref_ctx.sync()
            assert np.array_equal(h.bits(drvs[0].texture("Result")), h.bits(ref.texture("Result"))), frame
            for r in range(world):
                lo, hi = cuts[r], cuts[r + 1]
                assert np.array_equal(h.bits(drvs[r].texture("AccumulatedResult")[lo:hi]), h.bits(ref.texture("AccumulatedResult")[lo:hi])), (frame, r)
                assert np.array_equal(h.bits(drvs[r].buffer("Candidates")[lo * W * tot:hi * W * tot]),
                                      h.bits(ref.buffer("Candidates")[lo * W * tot:hi * W * tot])), (frame, r)
                assert np.array_equal(h.bits(drvs[r].buffer("HitInfos")[lo * W:hi * W]), h.bits(ref.buffer("HitInfos")[lo * W:hi * W])), (frame, r)
        assert ref.texture("Result")[..., :3].any()
    finally:
        for g in ranks:
            host.rth_mg_destroy(g)
        for d in drvs:
            d.close()
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("world,res,frames,timed,in_flight,delay_us", [(2, (128, 96), 5, False, 2, 0), (4, (256, 416), 7, False, 3, 0),
                                                                      (3, (96, 160), 4, True, 2, 0), (2, (160, 200), 9, False, 4, 0),
                                                                      (3, (128, 264), 8, False, 2, 150)])
def test_native_frame_loop_rehearsed_with_threads(scene_small, world, res, frames, timed, in_flight, delay_us):
    """rth_mg_render_frames itself with several ranks, on the one GPU: every rank is a thread with a context of its own,
    and the transfers the loop hands to RCCL go through the rehearsal hub instead (same matching and stream ordering,
    device copies).  All frames are enqueued without a wait in between - two to four frames in flight per rank, the gather of a
    frame behind the next frame's halo exchange - with a camera that moves from frame to frame, and a second call
    continues the sequence.  Rank 0's gathered image and every rank's band must equal the single-GPU render; and every
    rank's log of the second call (rt_debug_trace: launches, the loop's transfers with their rows, event records and
    waits) must show every two operations that touch the same rows of the same object, one of them writing, ordered
    by streams and events alone (harness.launch_order_violations) - whatever the timing was."""
    import ctypes as C
    import threading
    import harness as h
    rt = h.rt
    W, H = res
    cfg = h.small_config(W, H)
    host = rt.host_lib()
    ctxs = [rt.RtContext(0) for _ in range(world + 1)]
    for c in ctxs[:world]:
        c.set_tuning("frames_in_flight", in_flight)
        c.set_tuning("debug_stream_delay_us", delay_us)   # (test aid: the library's streams held up at random, see test_gpu_overlap)
    drvs = [rt.Driver(c, cfg, scene_small) for c in ctxs]
    ref_ctx, ref = ctxs[world], drvs[world]
    hub = host.rth_hub_create(world)
    assert hub
    ranks = []
    err = C.create_string_buffer(512)
    try:
        for r in range(world):
            g = host.rth_mg_create(drvs[r].h, r, world, None, None, err, 512)
            assert g, err.value.decode()
            ranks.append(g)
            assert host.rth_mg_use_hub(g, hub) == 0, host.rth_mg_last_error(g)
            if timed:
                assert host.rth_mg_enable_timing(g, 1) == 0
        b = (C.c_int * (world + 1))()
        host.rth_mg_bounds(ranks[0], b)
        cuts = list(b)

        def run(n):
            failures = []

            def work(g):
                if host.rth_mg_render_frames(g, n) != 0:
                    failures.append(host.rth_mg_last_error(g))

            threads = [threading.Thread(target=work, args=(g,)) for g in ranks]
            for t in threads:
                t.start()
            for t in threads:
                t.join(120)
            assert not failures and not any(t.is_alive() for t in threads), failures

        def check(what):
            for c in ctxs[:world]:
                c.sync()
            ref_ctx.sync()
            assert np.array_equal(h.bits(drvs[0].texture("Result")), h.bits(ref.texture("Result"))), what
            for r in range(world):
                lo, hi = cuts[r], cuts[r + 1]
                for name in ("AccumulatedResult", "DepthBuffer"):
                    assert np.array_equal(h.bits(drvs[r].texture(name)[lo:hi]), h.bits(ref.texture(name)[lo:hi])), (what, r, name)
                for name in ("Candidates", "TemporalFrameBuffer", "HitInfos"):
                    assert np.array_equal(h.bits(drvs[r].buffer(name)[lo * W:hi * W]), h.bits(ref.buffer(name)[lo * W:hi * W])), (what, r, name)

        pos = np.array(cfg.cameraPosition[:], np.float32)
        rot = [float(x) for x in cfg.cameraRotation]
        # (the camera is set once per call: the native loop renders its n frames from one camera, as Main.Update does
        # between two camera moves)
        for part, n in enumerate((frames, 3)):
            p = pos + np.float32(0.004 * part) * np.array([1, 0, 0.5], np.float32)
            for d in drvs:
                d.set_camera(p, h.quat((0.2, 1, 0), 7.0 * part, rot))
            if part == 1:      # the second call is also logged: launches, transfers, event records and waits of every rank
                for c in ctxs[:world]:
                    c.sync()
                    c.trace(True)
            run(n)
            for _ in range(n):
                ref.render_frame()
            check(f"call {part}")
            if part == 1:
                for r, c in enumerate(ctxs[:world]):
                    c.trace(False)
                    text = c.trace_text()
                    assert "transfer" in text
                    bad = h.launch_order_violations(text)
                    assert not bad, f"rank {r}:\n" + "\n".join(bad)
    finally:
        for g in ranks:
            host.rth_mg_destroy(g)
        host.rth_hub_destroy(hub)
        for d in drvs:
            d.close()
        for c in ctxs:
            c.close()

