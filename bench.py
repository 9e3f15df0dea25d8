#!/usr/bin/env python3
"""Benchmark of the ReSTIR direct-lighting frame on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one frame of the reference's sequence (InitialTrace, TemporalReuse, 3 x SpatialReusePass,
TransferToOriginal, TraceRays, AccumulateFrames) on the "80kScene*" stand-in at 1920x1080, all inputs resident in
HBM before the timed region.  N > 1 splits the same image into N row bands, one process per GPU (launched by
torch.distributed.run), with reservoir halos exchanged between neighbours once per frame and the displayed
image gathered on rank 0 (strong scaling).  Rank 0 prints ONE JSON line.

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


def pmc_traffic(kernel, W, H, world, small_scene, scene="80k"):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE and WRITE_SIZE, one rocprofv3 --pmc
    run each over this same command; tools/pmc_pass.sh, tools/pmc_to_traffic.py), or None when this run is not the
    configuration they were collected on.  Counters cannot be read from inside the process being timed."""
    if (W, H, world, small_scene, scene) != (1920, 1080, 1, False, "80k"):
        return None
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if not os.path.exists(path):
        return None
    k = json.load(open(path))["kernels"].get(kernel)
    return k["hbm_bytes_per_launch"] if k else None


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
    ap.add_argument("--cpu-frames", type=int, default=2)
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
    ap.add_argument("--exchange", choices=("rccl", "torch"), default="rccl",
                    help="N > 1: halo exchange from the C++ frame loop with RCCL calls (default), or from Python with "
                         "torch.distributed's NCCL backend")
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
    # Warm-up, with every launch bracketed by events: tells which kernel dominates the frame.
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(args.warmup):
        runner.render_frame()
    runner.sync()
    wprof = ctx.profile()
    dominant = max(wprof, key=lambda k: wprof[k]["ms"]) if wprof else ("ris_trace_rays" if ris else "trace_rays")
    # Timed region: the kernels count their rays, and only the dominant kernel's launches carry events (a pair of
    # events costs a few microseconds of stream time; around all ten launches of a frame they cost 6 % of it).
    ctx.profile_reset()
    ctx.profile_enable(True, only=dominant)
    runner.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.render_frame()
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
    for _ in range(n_breakdown):
        runner.render_frame()
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
    for _ in range(n_instr):
        runner.render_frame()
    runner.sync()
    iprof = ctx.profile()
    ctx.enable_stats(0)

    if rank == 0:
        proj = "BVH_RIS_DL" if ris else "BVH_ReSTIR_DL"
        band_px = runner.band_pixels()
        kernels = {}
        for name, p in prof.items():
            ms = p["ms"] / max(1, p["launches"])
            entry = {"launches_per_frame": p["launches"] / steps_of[name], "avg_ms": ms}
            if name in iprof:
                c = {k: v / iprof[name]["launches"] for k, v in iprof[name].items() if k not in ("launches", "ms")}
                b = algorithmic_bytes(name, c, band_px)
                entry["counters_per_launch"] = {k: round(v) for k, v in c.items() if v}
                entry["algorithmic_MB_per_launch"] = b / 1e6
                entry["achieved_GBs"] = b / 1e9 / (ms / 1e3) if ms > 0 else None
            kernels[name] = entry
        roofline = None
        if kernels.get(dominant, {}).get("achieved_GBs"):
            d = kernels[dominant]
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": d["achieved_GBs"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": d["achieved_GBs"] / HBM_PEAK_GBS,
                        "traffic": None if ris else pmc_traffic(dominant, W, H, world, args.small_scene, args.scene),
                        "avg_launch_ms": d["avg_ms"], "algorithmic_MB_per_launch": d["algorithmic_MB_per_launch"]}
            if roofline["traffic"]:   # what actually crossed the HBM interface, as a rate
                roofline["hbm_GBs_from_pmc"] = roofline["traffic"] / 1e9 / (d["avg_ms"] / 1e3)
        # this rank's rays over the time of the kernels that cast them (InitialTrace + TraceRays) only
        trav_ms = sum(v["avg_ms"] * v["launches_per_frame"] for k, v in kernels.items()
                      if k in ("primary", "candidates", "trace_rays", "ris_primary", "ris_trace_rays"))
        trav_rate = rays_local / args.steps / 1e6 / (trav_ms / 1e3) if trav_ms > 0 else None
        out = {
            "metric": "Mrays/s (primary+shadow) " + ("RIS DL" if ris else "ReSTIR DL") + ", 80kScene 1080p",
            "value": rays_total / 1e6 / dt_max, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{proj} 80kScene* ({'8.7k' if args.small_scene else '78k'}-triangle dragon stand-in "
                                    f"+ Suzanne light)" if args.scene == "80k" else
                                    f"{proj} {args.scene} 80kScene* (128 dragon "
                                    f"{'instances' if args.scene == 'instanced' else 'meshes, each distinct'} + 10 emissive discs{' (fans)' if args.emissive_fans else ''})") +
                                   (f" {W}x{H}, RIS project: {RIS_TOT} candidates per pixel, {RIS_TESTS} drawn" if ris else
                                    f" {W}x{H}, temporal + 3 spatial reuse passes, VIS_REUSE+BRDF"),
                       "instanced_triangles": int(sum(scene.arrays()["ComponentDatas"][o["bvStartIndex"]][1]
                                                      for o in scene.arrays()["SceneObjects"])),
                       "resolution": [W, H], "triangles": int(len(scene.arrays()['Triangles'])),
                       "parallelism": f"row-bands x{world} ({args.exchange} halo exchange)" if world > 1 else "single GPU",
                       "rays_per_frame": rays_total / args.steps},
            "roofline": roofline, "kernels": kernels,
            "traversal_kernels_Mrays_per_s": trav_rate,
            "kernel_timing": f"HIP events on the library's stream: '{dominant}' around every launch of the timed region, "
                             f"the other kernels in {n_breakdown} untimed frames after it",
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(rt, scene, cfg, args.cpu_frames, threads)
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
