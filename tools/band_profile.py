"""Per-kernel times when one GPU renders only a row band of the 1080p frame (what one rank of an N-GPU run does)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
cfg = rt.default_config()
for (r0, r1) in ((0, 0), (400, 672), (472, 608), (0, 136)):
    ctx = rt.RtContext(0)
    drv = rt.Driver(ctx, cfg, scene, r0, r1)
    for _ in range(8):
        drv.render_frame()
    ctx.sync()
    ctx.profile_reset(); ctx.profile_enable(True)
    n = 32
    t0 = time.perf_counter()
    for _ in range(n):
        drv.render_frame()
    ctx.sync()
    wall = (time.perf_counter() - t0) / n
    p = ctx.profile()
    tot = sum(v["ms"] for v in p.values()) / n
    print(f"rows [{r0},{r1}): {tot*1e3:.0f} us/frame in kernels, {wall*1e6:.0f} us wall  " + " ".join(f"{k}={v['ms']/v['launches']*1e3:.0f}" for k, v in p.items()))
    drv.close(); ctx.close()
