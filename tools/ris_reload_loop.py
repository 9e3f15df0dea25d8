import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
ctx = rt.RtContext(0)
rcfg = rt.ris_default_config()
rcfg.Resolution[0], rcfg.Resolution[1] = 1920, 1080
rcfg.TriCandidatesNum, rcfg.CandidateReservoirTestsNum = 32, 32
rdrv = rt.RisDriver(ctx, rcfg, scene)
for _ in range(4):
    rdrv.render_frame()
ctx.sync()
for k in range(12):
    scene.set_object_transform(1, (-3.0 + 0.01 * k, -2.0, 12.0), (0.0, 25.0, 0.0), (12.0, 12.0, 12.0))
    scene.rebuild(keep_old_scene_bvh=False)
    rdrv.reload()
    rdrv.render_frame()
ctx.sync()
rdrv.close(); ctx.close()
