"""Long run: thousands of frames with timing and counters on; device memory must not grow and the accumulated image must stay finite."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=False)
cfg = rt.default_config()
cfg.Resolution[0], cfg.Resolution[1] = 320, 240
ctx = rt.RtContext(0)
drv = rt.Driver(ctx, cfg, scene)
ctx.enable_stats(1)
ctx.profile_enable(True)
for _ in range(50):
    drv.render_frame()
ctx.sync()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
n = 5000
for f in range(n):
    drv.render_frame()
    if f % 1000 == 999:
        ctx.sync()
        print(f + 1, "frames", f"{(time.perf_counter() - t0) / (f + 1) * 1e6:.0f} us/frame", "free MB", torch.cuda.mem_get_info()[0] >> 20, flush=True)
ctx.sync()
free1 = torch.cuda.mem_get_info()[0]
acc = drv.texture("AccumulatedResult")
p = ctx.profile()
print("device memory change:", (free0 - free1) >> 10, "KiB; accumulated finite:", bool(np.isfinite(acc[..., :3]).all()),
      "launches", p["trace_rays"]["launches"])
assert free0 - free1 < (64 << 20)
