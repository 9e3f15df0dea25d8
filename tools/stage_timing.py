"""Diagnostics: per-kernel HIP-event times of the frame at stats level 0 and 1, and stage-by-stage host timing."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
cfg = rt.default_config()
if len(sys.argv) > 2:
    cfg.Resolution[0], cfg.Resolution[1] = int(sys.argv[1]), int(sys.argv[2])
ctx = rt.RtContext(0)
drv = rt.Driver(ctx, cfg, scene)
for level in (0, 1, 2):
    ctx.enable_stats(level)
    for _ in range(8):
        drv.render_frame()
    ctx.sync()
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    n = 64
    for _ in range(n):
        drv.render_frame()
    ctx.sync()
    dt = (time.perf_counter() - t0) / n * 1e3
    ctx.profile_enable(False)
    p = ctx.profile()
    print(f"stats level {level}: {dt:.3f} ms/frame; kernels sum {sum(v['ms'] for v in p.values()) / n:.3f} ms; " +
          " ".join(f"{k}={v['ms'] / v['launches'] * 1e3:.0f}us" for k, v in p.items()))
ctx.enable_stats(0)
ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(64):
    drv.render_frame()
ctx.sync()
print(f"no timing, no stats: {(time.perf_counter() - t0) / 64 * 1e3:.3f} ms/frame")
