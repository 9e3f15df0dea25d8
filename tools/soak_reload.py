"""Long run of scene reloads (an object moves before every frame; every tenth reload also changes a setting): the
device's free memory must settle - released memory is kept for the next reload, not piled up - and the frames at the
end must still be the oracle's."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as h  # noqa: E402

rt = h.rt
hip = C.CDLL("libamdhip64.so")


def free_mb():
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value >> 20


n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
scene = rt.build_scene_s0(subdivide=False)
cfg = h.small_config(320, 240)
ctx = rt.RtContext(0)
drv = rt.Driver(ctx, cfg, scene)
orc = h.OracleRenderer(scene.arrays(), cfg)
drv.render_frame(); orc.render_frame()
seen = []
for k in range(n):
    scene.set_object_transform(1, (-3.0 + 0.01 * (k % 50), -2.0, 12.0), (0.0, 25.0 + 0.5 * (k % 30), 0.0), (12.0, 12.0, 12.0))
    scene.rebuild()
    new_cfg = None
    if k % 10 == 9:
        new_cfg = h.small_config(320, 240)
        new_cfg.TriReservoirTestsNum = 5 + (k // 10) % 4
    drv.reload(new_cfg)
    drv.render_frame()
    orc.reload(scene.arrays(), new_cfg)   # (the oracle follows every frame: the accumulated image carries the history)
    orc.render_frame()
    if k % 50 == 49:
        ctx.sync()
        seen.append(free_mb())
        print(k + 1, "reloads, free MB", seen[-1], "BVs", len(scene.arrays()["BVs"]), flush=True)
ctx.sync()
for name in ("HitInfos", "Candidates", "TemporalFrameBuffer"):
    assert np.array_equal(h.bits(drv.buffer(name)), h.bits(orc.buf[name])), name
for name in ("Result", "AccumulatedResult", "DepthBuffer"):
    assert np.array_equal(h.bits(drv.texture(name)), h.bits(orc.tex[name])), name
print("frames after", n, "reloads equal the oracle's")
assert max(seen[1:]) - min(seen[1:]) <= 64, seen   # (BVs grows by three nodes per reload: a size class up now and then)
print("free memory settled:", seen)
print("ok")
drv.close()
ctx.close()
