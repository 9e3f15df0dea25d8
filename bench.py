#!/usr/bin/env python3
"""Benchmark of the ReSTIR direct-lighting frame on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one frame of the reference's sequence (InitialTrace, TemporalReuse, 3 x SpatialReusePass,
TransferToOriginal, TraceRays, AccumulateFrames) on the "80kScene*" stand-in at 1920x1080, all inputs resident in
HBM before the timed region.  N > 1 splits the same image into N row bands, one process per GPU (launched by
torch.distributed.run), with reservoir halos exchanged between neighbours once per frame and the displayed
image gathered on rank 0 (strong scaling).  Rank 0 prints ONE JSON line.

The headline run (no workload flags) also times the same frame at 3840x2160 on the same ranks, after everything
else, and reports it as `also_3840x2160`: the multi-GPU target is stated for that resolution.  `--no-4k` leaves it out.

value = rays cast by all ranks in the K timed frames / wall time of those frames, in Mrays/s (primary + shadow +
light rays, counted by the kernels themselves in the timed region).  The roofline figure is for the kernel that
dominates the frame: its launches are bracketed by HIP events inside the timed region; the per-kernel table of the
other kernels comes from a few untimed frames, because a pair of events around each of a frame's ten launches
lengthens the frame by 6 %.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
RIS_TOT = RIS_TESTS = 32  # --project ris: candidates stored per pixel, candidates drawn per pixel in TraceRays


def algorithmic_bytes(kernel, c, npx_launch):
    """Algorithmic bytes of ONE launch from its work counters `c` (per-launch averages).

    Traversal kernels: 32 B per node popped and per child box fetched, 168 B per scene-object visit, 28 B per
    leaf-loop triangle, 60 B (three vertices) per triangle tested — the reference's access pattern — plus the
    per-pixel records the kernel streams.  Reuse passes: their per-pixel record traffic.
    """
    c = {k: c.get(k, 0) for k in ("pops", "bvChecks", "objVisits", "triIters", "triChecks", "activePixels", "rays",
                                   "validNeighbours")}
    px = c["activePixels"]
    if kernel.startswith("ris_"):
        # RIS project: 76-byte triangles with inline vertices (one fetch per leaf-loop iteration), 40-byte HitInfo,
        # 16-byte candidates, RIS_TOT of them per pixel
        trav = 32 * (c["pops"] + c["bvChecks"]) + 168 * c["objVisits"] + 76 * c["triIters"]
        if kernel == "ris_primary":
            return trav + npx_launch * (40 + 16 + 16)       # HitInfo, DebugOverlay, RayHitPoints written
        if kernel == "ris_candidates":
            # the kernel counts the candidates it generated (RIS_TOT per hit pixel): 16 B written each, plus the
            # pixel's HitInfo read
            return c["validNeighbours"] * 16 + c["validNeighbours"] / RIS_TOT * 40
        if kernel == "ris_trace_rays":
            return trav + npx_launch * (40 + 16) + px * 16 * RIS_TESTS   # HitInfo, Result; candidates drawn
        return 0
    trav = 32 * (c["pops"] + c["bvChecks"]) + 168 * c["objVisits"] + 28 * c["triIters"] + 60 * c["triChecks"]
    if kernel == "primary":
        return trav + px * (52 + 16 + 4 + 16 + 16)          # HitInfo, DebugOverlay, depth, normals, hit point
    if kernel == "candidates":
        return trav + npx_launch * 52 + c["rays"] * 52      # HitInfo read per pixel, reservoir write per candidate
    if kernel == "trace_rays":
        return trav + px * (52 + 52 + 16)
    if kernel == "temporal":
        return px * (52 + 52 + 4 + 16 + 16 + 52 + 52)
    if kernel == "spatial":
        return px * (4 + 52 + 52) + c["validNeighbours"] * (4 + 52)
    if kernel == "transfer":
        return px * (4 + 52 + 52)
    if kernel == "accumulate":
        return npx_launch * 48
    return 0


def source_hash():
    """Hash of the sources the GPU library is built from: the committed PMC figures are only quoted for the build they
    were collected on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "ray-tracer_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_file(W, H, small_scene, scene, project, extra=None):
    """The committed PMC passes of one workload: profiles/r03_pmc_<project>_<scene>_<W>x<H>[...].json."""
    tag = f"{project}_{scene}{'_small' if small_scene else ''}_{W}x{H}"
    for k, v in sorted((extra or {}).items()):
        tag += f"_{k}{v}"
    return os.path.join(ROOT, "profiles", f"r03_pmc_{tag}.json")


def frames_in_flight_default(var="RT_FRAMES_IN_FLIGHT", default=2):
    """What new contexts start with (the environment variable, else the library's default: 2 for the ReSTIR project's
    frames, 3 for the RIS project's)."""
    e = os.environ.get(var, "")
    return max(1, min(4, int(e))) if e.isdigit() else default


def pmc_profile(W, H, world, small_scene, scene, project, extra=None):
    """(per-kernel PMC figures, None) from the committed rocprofv3 --pmc passes over this same command
    (tools/collect_profiles.sh -> tools/make_pmc_profile.py), or (None, reason) when they do not describe this run:
    another workload, or a library built from other sources than the ones the counters were collected on.
    Hardware counters cannot be read from inside the process being timed."""
    path = pmc_file(W, H, small_scene, scene, project, extra)
    if not os.path.exists(path):
        return None, f"no committed PMC passes for this workload ({os.path.relpath(path, ROOT)})"
    prof = json.load(open(path))
    want = {"width": W, "height": H, "world": world, "small_scene": small_scene, "scene": scene, "project": project}
    if extra:
        want.update(extra)
    if prof.get("config") != want:
        return None, f"the committed PMC passes are for {prof.get('config')}, this run is {want}"
    if prof.get("source_hash") != source_hash():
        return None, (f"the committed PMC passes were collected on sources {prof.get('source_hash')}, "
                      f"the library here is built from {source_hash()}")
    return prof["kernels"], None


def cpu_baseline(rt, scene, cfg, frames, threads):
    """The CPU oracle (a port of the reference's kernels) on the same scene, resolution and constants."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness as h
    ris = isinstance(cfg, rt.RisConfig)
    orc = (h.RisOracleRenderer if ris else h.OracleRenderer)(scene.arrays(), cfg, threads=threads, records=False)
    orc.render_frame()  # frame 0 (no temporal history yet): untimed warm-up
    orc.take_counters()
    t0 = time.perf_counter()
    for _ in range(frames):
        orc.render_frame()
    dt = time.perf_counter() - t0
    c = orc.take_counters()
    out = {"value": c["rays"] / 1e6 / dt, "unit": "Mrays/s", "cores": threads, "kind": "port",
           "sample": f"{frames} full {'RIS' if ris else 'ReSTIR'} frame(s) (frame index 1..{frames}) of the same "
                     f"{cfg.Resolution[0]}x{cfg.Resolution[1]} workload, "
                     f"{dt / frames * 1e3:.0f} ms/frame on {threads} threads",
           "ms_per_frame": dt / frames * 1e3}
    if threads > 1:   # one more frame on a single thread: the per-core figure
        orc.threads = 1
        t0 = time.perf_counter()
        orc.render_frame()
        dt1 = time.perf_counter() - t0
        out["single_thread"] = {"value": orc.take_counters()["rays"] / 1e6 / dt1, "unit": "Mrays/s",
                                "sample": f"1 further frame, {dt1 * 1e3:.0f} ms"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--small-scene", action="store_true", help="8.7k-triangle dragon instead of the 78k stand-in")
    ap.add_argument("--scene", choices=("80k", "instanced", "unique"), default="80k",
                    help="instanced: 16x8 instances of the 78k dragon (10 M triangles through one shared mesh BVH) lit by "
                         "10 discs of 1000 emissive triangles: the deep-BVH configuration.  unique: the same grid with "
                         "every dragon a distinct (vertex-jittered) mesh with its own BVH: 10 M stored triangles, "
                         "about 1.4 GB of nodes, triangles and vertices")
    ap.add_argument("--emissive-fans", action="store_true",
                    help="instanced / unique scenes: emissive discs as plain 1000-triangle fans instead of 40 sectors x 13 rings")
    ap.add_argument("--project", choices=("restir", "ris"), default="restir",
                    help="ris: the RIS project's frame (GenerateCandidates with 32 candidates per pixel, TraceRays "
                         "drawing 32 of them, AccumulateFrames) instead of the ReSTIR sequence")
    ap.add_argument("--tri-reservoir-tests", type=int, default=None,
                    help="ReSTIR project: TriReservoirTestsNum (default 5; the 256x256 CPU-runnable configuration uses 32)")
    ap.add_argument("--exchange", choices=("rccl", "torch"), default="rccl",
                    help="N > 1: halo exchange from the C++ frame loop with RCCL calls (default), or from Python with "
                         "torch.distributed's NCCL backend")
    ap.add_argument("--no-4k", action="store_true",
                    help="headline run only: leave out the extra measurement of the same frame at 3840x2160 (`also_3840x2160`)")
    ap.add_argument("--tuning", action="append", default=[], metavar="NAME=VALUE",
                    help="rt_set_tuning(NAME, VALUE) before the first frame (experiments; repeatable)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RT_BENCH_DEVICE" in os.environ:   # rehearsals on a box with fewer GPUs than ranks
        local_rank = int(os.environ["RT_BENCH_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    rt = importlib.import_module("ray-tracer_amd")
    if args.scene != "80k":
        scene = rt.build_scene_instanced(grid=(16, 8), emissive_objects=10, emissive_tris=1000,
                                         unique=args.scene == "unique", subdivide=not args.small_scene,
                                         emissive_fans=args.emissive_fans)
    else:
        scene = rt.build_scene_s0(subdivide=not args.small_scene)
    ris = args.project == "ris"
    if ris:
        cfg = rt.ris_default_config()
        cfg.TriCandidatesNum, cfg.CandidateReservoirTestsNum = RIS_TOT, RIS_TESTS
        args.exchange = "rccl"        # its banded loop exists in the native driver only
    else:
        cfg = rt.default_config()
        if args.tri_reservoir_tests is not None:
            cfg.TriReservoirTestsNum = args.tri_reservoir_tests
    cfg.Resolution[0], cfg.Resolution[1] = args.width, args.height
    W, H = args.width, args.height

    if world > 1:
        import torch
        import torch.distributed as dist
        mg = importlib.import_module("ray-tracer_amd.multigpu")
        runner = None
        if args.exchange == "rccl":
            # the side channel (unique id, timing reductions) first, so that the ranks can agree on whether the
            # native loop came up everywhere; if it did not on some rank, all of them take the torch.distributed path
            dist.init_process_group("gloo", rank=rank, world_size=world)
            why = ""
            try:
                runner = mg.RcclRunner(rt, scene, cfg, rank, world, local_rank)
            except Exception as e:  # noqa: BLE001 - any failure means "use the other driver"
                why = f"{type(e).__name__}: {e}"
            ok = torch.tensor([1 if runner else 0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0:
                if ris:
                    sys.exit(f"rank {rank}: the native multi-GPU loop failed to start ({why or 'on another rank'})")
                print(f"rank {rank}: native RCCL loop unavailable ({why or 'failed on another rank'}); "
                      f"using torch.distributed", file=sys.stderr)
                if runner:
                    runner.close()
                    runner = None
                elif dist.is_initialized():
                    dist.destroy_process_group()
                args.exchange = "torch"
        if runner is None:
            runner = mg.DistributedRunner(rt, scene, cfg, rank, world, local_rank)
    else:
        runner = SingleRunner(rt, scene, cfg)

    ctx = runner.ctx
    ctx.enable_stats(1)
    for item in args.tuning:
        name, _, value = item.partition("=")
        ctx.set_tuning(name, int(value))

    def render(n):
        # the native multi-GPU loop takes the whole run of frames in one call: it keeps the gather of a frame behind the
        # next frame's halo exchange, and completes the last frame's gather at the end of the call
        if hasattr(runner, "render_frames"):
            runner.render_frames(n)
        else:
            for _ in range(n):
                runner.render_frame()

    render(args.warmup)   # untimed: tables ranked, temporal history built, clocks up
    runner.sync()
    # Which kernel dominates the frame: the per-kernel table of 8 further untimed frames in steady state, every launch
    # bracketed by events - the kernel with the largest time per frame (average launch x launches per frame).  (Not taken
    # from the warm-up frames: with few of them the first, untypical frames decide.)  On one GPU these frames run with
    # every kernel ALONE on the chip (no primary rays beside the previous frame, one frame in flight): in the timed
    # region the library runs kernels of two frames beside one another, and the time between a launch's two events
    # then includes the share of the chip its neighbours took - good for the frame time, useless for telling how long
    # a kernel's own work takes.  The table of these frames is reported as `isolated_avg_ms`.
    isolated = world == 1
    if isolated:
        ctx.set_overlap(0)
        ctx.set_tuning("frames_in_flight", 1)
        ctx.set_tuning("ris_frames_in_flight", 1)
        render(2)
        runner.sync()
    ctx.profile_reset()
    ctx.profile_enable(True)
    render(8)
    runner.sync()
    wprof = ctx.profile()
    ctx.profile_enable(False)
    if isolated:
        ctx.set_overlap(1)
        ctx.set_tuning("frames_in_flight", frames_in_flight_default())
        ctx.set_tuning("ris_frames_in_flight", frames_in_flight_default("RT_RIS_FRAMES_IN_FLIGHT", 3))
        render(4)   # back into the steady state of the mode that is timed
        runner.sync()
    if os.environ.get("RT_BENCH_DEBUG"):
        print({k: round(v["ms"] / max(1, v["launches"]) * 1e3, 1) for k, v in wprof.items()}, file=sys.stderr)
    dominant = max(wprof, key=lambda k: wprof[k]["ms"]) if wprof else ("ris_trace_rays" if ris else "trace_rays")
    # Timed region: the kernels count their rays, and only the dominant kernel's launches carry events (a pair of
    # events costs a few microseconds of stream time; around all ten launches of a frame they cost 6 % of it).
    ctx.profile_reset()
    ctx.profile_enable(True, only=dominant)
    runner.barrier()
    t0 = time.perf_counter()
    render(args.steps)
    runner.sync()
    runner.barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    tprof = ctx.profile()
    rays_local = sum(k["rays"] for k in tprof.values())
    dt_max, rays_total = runner.reduce(dt, rays_local)

    # Per-kernel breakdown: the same frames' successors with events around every launch, untimed.
    ctx.profile_reset()
    ctx.profile_enable(True)
    n_breakdown = max(1, min(args.steps, 16))
    render(n_breakdown)
    runner.sync()
    ctx.profile_enable(False)
    prof = ctx.profile()
    if dominant in tprof and dominant in prof:   # the dominant kernel keeps its timed-region figure
        prof[dominant]["ms"], prof[dominant]["launches"] = tprof[dominant]["ms"], tprof[dominant]["launches"]
    steps_of = {name: (args.steps if name == dominant and dominant in tprof else n_breakdown) for name in prof}

    # Work counters for the roofline: a few more frames with the instrumented kernels, untimed.
    ctx.profile_reset()
    ctx.enable_stats(2)
    n_instr = 4
    render(n_instr)
    runner.sync()
    iprof = ctx.profile()
    ctx.enable_stats(0)

    # N > 1: where each rank's frame goes - dispatch-stream time before / beside / after the halo exchange, the exchange
    # and the gather on the transfer stream, and the part of the exchange the dispatch stream had to wait for.
    per_rank = None
    if world > 1 and hasattr(runner, "timed_frames"):
        import torch.distributed as dist
        mine = {"rank": rank, "rows": [int(runner.bounds[rank]), int(runner.bounds[rank + 1])],
                "ms": {k: round(v, 4) for k, v in runner.timed_frames(8).items()}}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # The headline run also measures the SAME frame at 3840x2160 on the same ranks - the configuration the multi-GPU
    # target is stated for (BASELINE.json configs[3]) - and reports it beside the 1080p figures (`also_3840x2160`):
    # a second runner of the same kind, its own warm-up, the same barrier / max-over-ranks timing.  It never changes
    # `value`; whatever goes wrong in it is reported in its place.
    also_4k = None
    headline = (not ris and args.scene == "80k" and (W, H) == (1920, 1080) and not args.small_scene
                and args.tri_reservoir_tests is None and not args.emissive_fans and not args.tuning)
    if headline and not args.no_4k and (world == 1 or type(runner).__name__ == "RcclRunner"):
        runner4 = None
        try:
            cfg4 = rt.default_config()
            cfg4.Resolution[0], cfg4.Resolution[1] = 3840, 2160
            if world == 1:
                runner4 = SingleRunner(rt, scene, cfg4)
            else:
                runner4 = mg.RcclRunner(rt, scene, cfg4, rank, world, local_rank)
            ctx4 = runner4.ctx
            ctx4.enable_stats(1)
            steps4 = max(1, min(args.steps, 24))

            def render4(n):
                if hasattr(runner4, "render_frames"):
                    runner4.render_frames(n)
                else:
                    for _ in range(n):
                        runner4.render_frame()

            render4(max(2, min(args.warmup, 8)))
            runner4.sync()
            ctx4.profile_reset()
            ctx4.profile_enable(True, only="trace_rays")
            runner4.barrier()
            t4 = time.perf_counter()
            render4(steps4)
            runner4.sync()
            runner4.barrier()
            dt4 = time.perf_counter() - t4
            ctx4.profile_enable(False)
            rays4 = sum(k["rays"] for k in ctx4.profile().values())
            dt4_max, rays4_total = runner4.reduce(dt4, rays4)
            also_4k = {"workload": "the same scene and settings at 3840x2160 (BASELINE.json configs[3])", "n_gpus": world,
                       "steps": steps4, "ms_per_step": dt4_max / steps4 * 1e3, "value": rays4_total / 1e6 / dt4_max, "unit": "Mrays/s",
                       "rays_per_frame": rays4_total / steps4}
            if world > 1:
                also_4k["rows_of_rank"] = [int(b) for b in runner4.bounds]
        except Exception as e:  # noqa: BLE001 - reported, never fatal for the headline line
            also_4k = {"error": f"{type(e).__name__}: {e}"}
        finally:
            if runner4 is not None:
                try:
                    if world > 1:
                        runner4.close(keep_group=True)
                    else:
                        runner4.close()
                except Exception:  # noqa: BLE001
                    pass

    if rank == 0:
        proj = "BVH_RIS_DL" if ris else "BVH_ReSTIR_DL"
        band_px = runner.band_pixels()
        kernels = {}
        for name, p in prof.items():
            ms = p["ms"] / max(1, p["launches"])
            entry = {"launches_per_frame": p["launches"] / steps_of[name], "avg_ms": ms}
            if isolated and name in wprof and wprof[name]["launches"]:
                entry["isolated_avg_ms"] = wprof[name]["ms"] / wprof[name]["launches"]
            if name in iprof:
                c = {k: v / iprof[name]["launches"] for k, v in iprof[name].items() if k not in ("launches", "ms")}
                b = algorithmic_bytes(name, c, band_px)
                entry["counters_per_launch"] = {k: round(v) for k, v in c.items() if v}
                entry["algorithmic_MB_per_launch"] = b / 1e6
                entry["achieved_GBs"] = b / 1e9 / (ms / 1e3) if ms > 0 else None   # algorithmic bytes over time
                if entry.get("isolated_avg_ms"):
                    entry["isolated_achieved_GBs"] = b / 1e9 / (entry["isolated_avg_ms"] / 1e3)
            kernels[name] = entry
        # What crossed the HBM interface, per kernel, from the committed PMC passes of this same command and build.
        pmc, pmc_reason = pmc_profile(W, H, world, args.small_scene, args.scene, args.project,
                                      {"tri_reservoir_tests": args.tri_reservoir_tests} if args.tri_reservoir_tests is not None else None)
        for name, entry in kernels.items():
            k = (pmc or {}).get(name)
            if k and entry["avg_ms"] > 0:
                entry["hbm_MB_per_launch"] = k["hbm_bytes_per_launch"] / 1e6
                entry["hbm_GBs"] = k["hbm_bytes_per_launch"] / 1e9 / (entry["avg_ms"] / 1e3)
                entry["hbm_frac"] = entry["hbm_GBs"] / HBM_PEAK_GBS
                for key in ("valu_busy_frac", "wave_wait_frac", "ta_busy_frac", "td_busy_frac", "l1_accesses_per_cu_clk", "l1_hit_frac", "l2_hit_frac",
                            "hbm_fetch_MB_per_launch", "hbm_write_MB_per_launch"):
                    if key in k:
                        entry[key] = round(k[key], 3)
        # The roofline of the dominant kernel is drawn against the HBM roof with the bytes that really crossed it
        # (`traffic`, PMC), so `frac` is a fraction of the peak.  The algorithmic bytes - what the REFERENCE's access
        # pattern moves for the same work (32 bytes per node popped and per child box, 28 + 60 per triangle ...) - are
        # reported beside it: the traversal kernels re-read a 15 MB scene through L1/L2, so they are a multiple of the
        # traffic and say how much re-reading the caches absorb, not how busy the DRAM is.
        roofline = None
        if dominant in kernels:
            d = kernels[dominant]
            traffic = (pmc or {}).get(dominant, {}).get("hbm_bytes_per_launch")
            achieved = traffic / 1e9 / (d["avg_ms"] / 1e3) if traffic and d["avg_ms"] > 0 else None
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic,
                        "avg_launch_ms": d["avg_ms"]}
            if d.get("isolated_avg_ms"):
                # the same launch with the chip to itself (see the comment at the selection of `dominant`): what the
                # kernel's own work takes; `achieved` above is over the time the launch spent SHARING the chip
                iso_ms = d["isolated_avg_ms"]
                roofline["isolated"] = {"avg_launch_ms": iso_ms, "achieved": traffic / 1e9 / (iso_ms / 1e3) if traffic else None,
                                        "frac": traffic / 1e9 / (iso_ms / 1e3) / HBM_PEAK_GBS if traffic else None,
                                        "algorithmic_GBs": d.get("isolated_achieved_GBs")}
            if traffic is None:
                roofline["traffic_reason"] = pmc_reason
            if d.get("algorithmic_MB_per_launch"):
                roofline["algorithmic"] = {"MB_per_launch": d["algorithmic_MB_per_launch"], "GBs": d["achieved_GBs"],
                                           "times_the_hbm_traffic": d["algorithmic_MB_per_launch"] * 1e6 / traffic if traffic else None,
                                           "served_by": "L1 / L2 / Infinity Cache: what the traversal re-reads of the scene never reaches the DRAM"}
            if "valu_busy_frac" in d:   # no unit is saturated: the kernel is bound by its chains of dependent steps
                roofline["limiter"] = {"kind": "latency of dependent traversal steps", "valu_busy_frac": d["valu_busy_frac"],
                                       "wave_wait_frac": d["wave_wait_frac"], "ta_busy_frac": d["ta_busy_frac"],
                                       "td_busy_frac": d["td_busy_frac"], "l1_accesses_per_cu_clk": d["l1_accesses_per_cu_clk"],
                                       "l1_hit_frac": d["l1_hit_frac"], "l2_hit_frac": d.get("l2_hit_frac")}
        # this rank's rays over the time of the kernels that cast them (InitialTrace + TraceRays) only
        trav_ms = sum(v["avg_ms"] * v["launches_per_frame"] for k, v in kernels.items()
                      if k in ("primary", "candidates", "trace_rays", "ris_primary", "ris_trace_rays"))
        trav_rate = rays_local / args.steps / 1e6 / (trav_ms / 1e3) if trav_ms > 0 else None
        out = {
            "metric": "Mrays/s (primary+shadow) " + ("RIS DL" if ris else "ReSTIR DL") + ", " +
                      {"80k": "80kScene", "instanced": "instanced 10M-triangle 80kScene", "unique": "10M-triangle 80kScene (distinct meshes)"}[args.scene] +
                      (" (8.7k-triangle dragon)" if args.small_scene else "") +
                      (" 1080p" if (W, H) == (1920, 1080) else " 4K" if (W, H) == (3840, 2160) else f" {W}x{H}"),
            "value": rays_total / 1e6 / dt_max, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{proj} 80kScene* ({'8.7k' if args.small_scene else '78k'}-triangle dragon stand-in "
                                    f"+ Suzanne light)" if args.scene == "80k" else
                                    f"{proj} {args.scene} 80kScene* (128 dragon "
                                    f"{'instances' if args.scene == 'instanced' else 'meshes, each distinct'} + 10 emissive discs{' (fans)' if args.emissive_fans else ''})") +
                                   (f" {W}x{H}, RIS project: {RIS_TOT} candidates per pixel, {RIS_TESTS} drawn" if ris else
                                    f" {W}x{H}, temporal + 3 spatial reuse passes, VIS_REUSE+BRDF" +
                                    (f", TriReservoirTestsNum {args.tri_reservoir_tests}" if args.tri_reservoir_tests is not None else "")),
                       "instanced_triangles": int(sum(scene.arrays()["ComponentDatas"][o["bvStartIndex"]][1]
                                                      for o in scene.arrays()["SceneObjects"])),
                       "resolution": [W, H], "triangles": int(len(scene.arrays()['Triangles'])),
                       "parallelism": f"row-bands x{world} ({args.exchange} halo exchange)" if world > 1 else "single GPU",
                       "rays_per_frame": rays_total / args.steps},
            "roofline": roofline, "kernels": kernels,
            "traversal_kernels_Mrays_per_s": trav_rate,
            "kernel_timing": f"HIP events on the stream each launch goes to: '{dominant}' around every launch of the timed region, "
                             f"the other kernels in {n_breakdown} untimed frames after it" +
                             ("; isolated_avg_ms: 8 untimed frames before it with one frame in flight and no overlapped primary rays" if isolated else ""),
        }
        if also_4k:
            out["also_3840x2160"] = also_4k
        if per_rank:
            out["per_rank_frame_ms"] = per_rank
            out["per_rank_note"] = ("8 further frames, each waited for; 'frame' is dispatch-stream time, 'halo_exchange' and "
                                    "'gather' run on the transfer stream, 'halo_exposed' is the wait the exchange cost the frame")
        if world == 1 and not args.no_cpu_baseline:
            # all the host cores this process may use: its CPU affinity, or the share the launcher announces through
            # OMP_NUM_THREADS (the GPU boxes give a one-GPU job 16 of the machine's cores and say so there)
            threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            if os.environ.get("OMP_NUM_THREADS", "").isdigit():
                threads = max(1, min(threads, int(os.environ["OMP_NUM_THREADS"])))
            out["cpu_baseline"] = cpu_baseline(rt, scene, cfg, args.cpu_frames, threads)
            out["cpu_baseline"]["host_cpus_visible"] = os.cpu_count()
        print(json.dumps(out))
    runner.close()


class SingleRunner:
    def __init__(self, rt, scene, cfg):
        self.ctx = rt.RtContext(0)
        self.drv = (rt.RisDriver if isinstance(cfg, rt.RisConfig) else rt.Driver)(self.ctx, cfg, scene)
        self.npx = cfg.Resolution[0] * cfg.Resolution[1]

    def render_frame(self):
        self.drv.render_frame()

    def sync(self):
        self.ctx.sync()

    def barrier(self):
        pass

    def reduce(self, dt, rays):
        return dt, rays

    def band_pixels(self):
        return self.npx

    def close(self):
        self.drv.close()
        self.ctx.close()


if __name__ == "__main__":
    main()
