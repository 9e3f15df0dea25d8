"""Long run of the merged-dispatch path against the separate one: 600 frames each at 320x240, every 100th frame the
reservoirs, the result and the accumulated image must be bit-identical."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=False)
cfg = rt.default_config()
cfg.Resolution[0], cfg.Resolution[1] = 320, 240
ctx = rt.RtContext(0)
a, b = rt.Driver(ctx, cfg, scene), rt.Driver(ctx, cfg, scene)


def separate(d):
    d.update_per_frame()
    for f in (lambda: d.run_stage(0), lambda: d.run_stage(1), lambda: d.run_stage(2), lambda: d.run_stage(3), d.run_post_processing):
        f()
        ctx.sync()


for f in range(600):
    a.render_frame()
    separate(b)
    if f % 100 == 99:
        ctx.sync()
        for name in ("Candidates", "TemporalFrameBuffer", "HitInfos"):
            assert np.array_equal(a.buffer(name).view(np.uint8), b.buffer(name).view(np.uint8)), (f, name)
        for name in ("Result", "AccumulatedResult"):
            assert np.array_equal(a.texture(name).view(np.uint32), b.texture(name).view(np.uint32)), (f, name)
        print(f + 1, "frames identical", flush=True)
print("ok")
