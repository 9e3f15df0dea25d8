"""What the candidate kernel costs without its shadow ray, and with fewer light-sampling draws."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
for label, edit in [("default", {}), ("no visibility reuse", {"DoVisibilityReuse": 0}),
                    ("no visibility reuse, 1 draw each", {"DoVisibilityReuse": 0, "SceneObjectReservoirTestsNum": 1, "TriReservoirTestsNum": 1}),
                    ("32 triangle draws", {"TriReservoirTestsNum": 32})]:
    cfg = rt.default_config()
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
    print(label, " ".join(f"{k}={v['ms'] / v['launches'] * 1e3:.0f}us" for k, v in p.items()))
    drv.close(); ctx.close()
