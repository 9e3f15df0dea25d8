"""How much of a frame's time is idle chip?  Two (or more) independent frame loops on streams of their own, against one."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
cfg = rt.default_config()
cfg.Resolution[0], cfg.Resolution[1] = 1920, 1080
for n in (1, 2, 3, 4):
    ctxs = [rt.RtContext(0) for _ in range(n)]
    drvs = [rt.Driver(c, cfg, scene) for c in ctxs]
    for _ in range(8):
        for d in drvs:
            d.render_frame()
    for c in ctxs:
        c.sync()
    frames = 64
    t0 = time.perf_counter()
    for _ in range(frames):
        for d in drvs:
            d.render_frame()
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t0
    print(f"{n} concurrent frame loops: {dt / frames * 1e3:.3f} ms per round of {n} frames = {dt / frames / n * 1e3:.3f} ms per frame", flush=True)
    for d in drvs:
        d.close()
    for c in ctxs:
        c.close()
