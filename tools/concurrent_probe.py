"""How much of a frame's time is idle chip?  Two (or more) independent frame loops on streams of their own, against one.
usage: python tools/concurrent_probe.py [width height [row0 row1]]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
cfg = rt.default_config()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
r0, r1 = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
cfg.Resolution[0], cfg.Resolution[1] = W, H
for n in (1, 2, 3):
    ctxs = [rt.RtContext(0) for _ in range(n)]
    drvs = [rt.Driver(c, cfg, scene, r0, r1) for c in ctxs]
    if r1 > 0:
        for c in ctxs:
            c.set_overlap(2)
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
    print(f"{W}x{H} rows [{r0},{r1}): {n} concurrent frame loops: {dt / frames * 1e3:.3f} ms per round of {n} frames = {dt / frames / n * 1e3:.3f} ms per frame", flush=True)
    for d in drvs:
        d.close()
    for c in ctxs:
        c.close()
