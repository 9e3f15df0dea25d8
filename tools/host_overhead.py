"""Host time needed to ENQUEUE one frame (no waiting for the GPU) versus the GPU time of the frame, per resolution."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
for (w, h) in ((1920, 1080), (1920, 136), (256, 256)):
    cfg = rt.default_config()
    cfg.Resolution[0], cfg.Resolution[1] = w, h
    ctx = rt.RtContext(0)
    drv = rt.Driver(ctx, cfg, scene)
    for _ in range(10):
        drv.render_frame()
    ctx.sync()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        drv.render_frame()
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"{w}x{h}: enqueue {1e6 * (t1 - t0) / n:.1f} us/frame, total {1e6 * (t2 - t0) / n:.1f} us/frame")
    drv.close()
    ctx.close()
