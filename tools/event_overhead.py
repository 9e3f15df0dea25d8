"""Frame time with and without the library's per-launch HIP events (what does the per-kernel timing cost?)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
cfg = rt.default_config()
ctx = rt.RtContext(0)
drv = rt.Driver(ctx, cfg, scene)
for _ in range(8):
    drv.render_frame()
ctx.sync()
for label, timing, stats in (("events+counters", True, 1), ("counters only", False, 1), ("neither", False, 0), ("events+counters", True, 1), ("neither", False, 0)):
    ctx.enable_stats(stats)
    ctx.profile_reset(); ctx.profile_enable(timing)
    for _ in range(4):
        drv.render_frame()
    ctx.sync()
    t0 = time.perf_counter()
    n = 64
    for _ in range(n):
        drv.render_frame()
    ctx.sync()
    print(f"{label}: {(time.perf_counter() - t0) / n * 1e6:.1f} us/frame")
    ctx.profile_enable(False)
