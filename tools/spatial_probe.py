"""Time of the spatial reuse pass under variations of its constants (what bounds it?)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
for label, edit in [("default", {}), ("K=0", {"TemporalCandidatesNum": 0}), ("K=1", {"TemporalCandidatesNum": 1}),
                    ("K=4", {"TemporalCandidatesNum": 4}), ("no angle test", {"SpatialNormalsAngleThreshold": 200.0}),
                    ("blur 1", {"SpatialReuseBlur": 1})]:
    cfg = rt.default_config()
    cfg.Resolution[0], cfg.Resolution[1] = 1920, 1080
    for k, v in edit.items():
        setattr(cfg, k, v)
    ctx = rt.RtContext(0)
    drv = rt.Driver(ctx, cfg, scene)
    ctx.enable_stats(1)
    for _ in range(4):
        drv.render_frame()
    ctx.sync()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(16):
        drv.render_frame()
    ctx.sync()
    p = ctx.profile()
    print(label, " ".join(f"{k}={v['ms'] / v['launches'] * 1e3:.0f}us" for k, v in p.items() if k in ("spatial", "temporal", "transfer")))
    drv.close(); ctx.close()
