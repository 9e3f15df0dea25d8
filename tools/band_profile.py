"""What ONE rank of an N-GPU run does, measured on ONE GPU: not a scaling curve.

All N ranks of the banded frame loop are held in this process on the one GPU (as tests/test_gpu_multiprocess.py holds
them), with cost-balanced cuts; the transfers the loop would hand to RCCL are carried out with device copies.  For every
rank, alone on the GPU:

  phases  dispatch-stream time of the four steps of its frame (rt_mark stamps), each step waited for;
  frame   wall time per frame when the rank's steps are enqueued back to back for a run of frames - the way
          rth_mg_render_frames enqueues them - with the halos left as the last exchange put them (their content does
          not change what a frame costs), once as the loop runs them (rt_set_overlap 2: the next frame's primary rays
          and, on a second dispatch stream, the next frame's other kernels beside the previous frame's) and once with
          every kernel after the one before (rt_set_overlap 0); and the host time it takes to enqueue such a frame.

    python tools/band_profile.py [--width 3840 --height 2160 --ranks 8 --frames 24] > profiles/r03_band_times_4k.json
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
bindings = importlib.import_module("ray-tracer_amd.bindings")
mg = importlib.import_module("ray-tracer_amd.multigpu")

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--scene", choices=("80k", "instanced", "unique"), default="80k")
ap.add_argument("--refine", type=int, default=0,
                help="cut again this many times with every band's row costs scaled to the frame time measured for it")
args = ap.parse_args()
W, H, N = args.width, args.height, args.ranks

scene = rt.build_scene_s0(subdivide=True) if args.scene == "80k" else rt.build_scene_instanced(unique=args.scene == "unique")
cfg = rt.default_config()
cfg.Resolution[0], cfg.Resolution[1] = W, H
host = rt.host_lib()
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def timed_frames(ctx, drv, frames):
    for _ in range(4):
        drv.render_frame()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        drv.render_frame()
    ctx.sync()
    return (time.perf_counter() - t0) / frames * 1e3


# the whole frame on the one GPU, and the depth buffer the cuts are balanced on
ctx = rt.RtContext(0)
drv = rt.Driver(ctx, cfg, scene)
full_ms = timed_frames(ctx, drv, args.frames)
ctx.set_overlap(0)
full_serial_ms = timed_frames(ctx, drv, args.frames)
depth = drv.texture("DepthBuffer")
drv.close()
ctx.close()
_, _, total = mg.native_halo_plan(cfg.SpatialReuseIterations, cfg.SpatialReuseBlur)
cuts = mg.band_bounds(H, N, mg.row_cost(depth), total)

def measure(cuts):
    ctxs = [rt.RtContext(0) for _ in range(N)]
    drvs = [rt.Driver(c, cfg, scene) for c in ctxs]
    err = C.create_string_buffer(512)
    ranks = []
    for r in range(N):
        g = host.rth_mg_create(drvs[r].h, r, N, (C.c_int * (N + 1))(*cuts), None, err, 512)
        assert g, err.value.decode()
        ranks.append(g)


    def plan(which):
        out = []
        for g in ranks:
            buf = (bindings.Xfer * 64)()
            n = host.rth_mg_plan(g, which, buf, 64)
            assert n >= 0
            out.append([buf[i] for i in range(n)])
        return out


    def carry_out(plans):
        for c in ctxs:
            c.sync()
        for a in range(N):
            for b in range(N):
                sends = [x for x in plans[a] if x.send and x.peer == b]
                recvs = [x for x in plans[b] if not x.send and x.peer == a]
                for s, d in zip(sends, recvs):
                    assert hip.hipMemcpy(d.ptr, s.ptr, s.bytes, 3) == 0


    def elapsed(c, a, b):
        ms = C.c_double()
        c.check(c.lib.rt_mark_elapsed(c.h, a, b, C.byref(ms)), "rt_mark_elapsed")
        return ms.value


    # per-phase times: every rank in turn, each phase waited for; the halos really exchanged
    phase_ms = np.zeros((N, 4))
    rounds = 6
    for it in range(2 + rounds):
        for k in range(4):
            for r, g in enumerate(ranks):
                c = ctxs[r]
                c.sync()
                c.check(c.lib.rt_mark(c.h, 0, 0), "rt_mark")
                assert host.rth_mg_phase(g, k) == 0, host.rth_mg_last_error(g)
                c.lib.rt_flush(c.h)
                c.check(c.lib.rt_mark(c.h, 0, 1), "rt_mark")
                c.sync()
                if it >= 2:
                    phase_ms[r, k] += elapsed(c, 0, 1) / rounds
            if k == 0:
                halo = plan(0)
            if k == 1:
                carry_out(halo)

    # a rank's frames back to back, as the native loop enqueues them
    frame_ms = np.zeros((N, 2))
    enqueue_ms = np.zeros(N)
    for r, g in enumerate(ranks):
        c = ctxs[r]
        for j, mode in enumerate((2, 0)):
            c.set_overlap(mode)
            for _ in range(3):
                for k in range(4):
                    assert host.rth_mg_phase(g, k) == 0
            c.sync()
            t0 = time.perf_counter()
            for _ in range(args.frames):
                for k in range(4):
                    assert host.rth_mg_phase(g, k) == 0
            if j == 0:
                enqueue_ms[r] = (time.perf_counter() - t0) / args.frames * 1e3
            c.sync()
            frame_ms[r, j] = (time.perf_counter() - t0) / args.frames * 1e3
        c.set_overlap(2)

    for g in ranks:
        host.rth_mg_destroy(g)
    for d in drvs:
        d.close()
    for c in ctxs:
        c.close()
    return phase_ms, frame_ms, enqueue_ms


weights = mg.row_cost(depth)
history = []
phase_ms, frame_ms, enqueue_ms = measure(cuts)
for it in range(args.refine):
    # every band's rows scaled so that the band's total is the frame time measured for it; then cut again
    history.append({"cuts": [int(c) for c in cuts], "frame_ms_two_frames_in_flight": [float(x) for x in frame_ms[:, 0]],
                    "frame_ms_kernels_one_after_the_other": [float(x) for x in frame_ms[:, 1]]})
    scaled = weights.copy()
    for r in range(N):
        scaled[cuts[r]:cuts[r + 1]] *= frame_ms[r, 1] / weights[cuts[r]:cuts[r + 1]].sum()
    weights = scaled
    cuts = mg.band_bounds(H, N, weights, total)
    phase_ms, frame_ms, enqueue_ms = measure(cuts)

out = {
    "what": "one GPU, not a scaling curve: every rank of the banded frame loop run alone on the same MI355X",
    "workload": f"BVH_ReSTIR_DL {args.scene} scene {W}x{H}, {N} row bands with cost-balanced cuts, temporal + 3 spatial passes",
    "cuts": cuts, "halo_rows_exchanged": total,
    "full_frame_ms": {"default_two_frames_in_flight": full_ms, "kernels_one_after_the_other": full_serial_ms},
    "ranks": [{"rank": r, "rows": [cuts[r], cuts[r + 1]],
               "phase_ms": {"initial_trace+temporal": phase_ms[r, 0], "spatial_pass0_interior": phase_ms[r, 1],
                            "spatial_rest+transfer": phase_ms[r, 2], "trace_rays+accumulate": phase_ms[r, 3],
                            "sum": float(phase_ms[r].sum())},
               "frame_ms_back_to_back": {"two_frames_in_flight": frame_ms[r, 0], "kernels_one_after_the_other": frame_ms[r, 1]},
               "host_ms_to_enqueue_a_frame": enqueue_ms[r]} for r in range(N)],
}
worst = max(out["ranks"], key=lambda x: x["frame_ms_back_to_back"]["two_frames_in_flight"])
out["max_band_frame_ms"] = worst["frame_ms_back_to_back"]["two_frames_in_flight"]
out["max_band_over_full_frame"] = out["max_band_frame_ms"] / full_ms
out["speedup_if_transfers_were_free"] = full_ms / out["max_band_frame_ms"]
if history:
    out["cuts_before_refinement"] = history
    out["refinement"] = ("each earlier set of cuts and the band times measured with it; the next cuts balance the row costs "
                         "(1 per pixel + 24 per hit pixel) scaled per band to the band's frame time with its kernels one after the other")
print(json.dumps(out, indent=1))
