"""The same question as tools/concurrent_probe.py for the RIS project's frame: one, two and three independent frame
loops on one GPU (the bound for running consecutive RIS frames beside one another).
usage: python tools/concurrent_probe_ris.py [width height]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
cfg = rt.ris_default_config()
cfg.TriCandidatesNum, cfg.CandidateReservoirTestsNum = 32, 32
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
cfg.Resolution[0], cfg.Resolution[1] = W, H
for n in (1, 2, 3):
    ctxs = [rt.RtContext(0) for _ in range(n)]
    drvs = [rt.RisDriver(c, cfg, scene) for c in ctxs]
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
    print(f"RIS {W}x{H}: {n} concurrent frame loops: {dt / frames / n * 1e3:.3f} ms per frame", flush=True)
    for d in drvs:
        d.close()
    for c in ctxs:
        c.close()
