"""Diagnostics: per-wave lifetime and loop-iteration distribution of the primary-ray kernel (instrumented variant)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("ray-tracer_amd")
scene = rt.build_scene_s0(subdivide=True)
cfg = rt.default_config()
ctx = rt.RtContext(0)
drv = rt.Driver(ctx, cfg, scene)
for _ in range(3):
    drv.render_frame()
nb = 8192 * 2
which = int(os.environ.get("RT_WAVE_PROFILE_KERNEL", "0"))
wp = ctx.create_buffer(nb * 4 + 64, 64)
ctx.check(ctx.lib.rt_set_buffer(drv.shader("rt"), 4 if which == 2 else 0, b"WaveProfile", wp), "bind")
drv.update_per_frame()
for st in range(4):
    drv.run_stage(st)
ctx.sync()
a = ctx.get_buffer(wp, np.dtype([("t0", "<u8"), ("t1", "<u8"), ("mx", "<u8"), ("sm", "<u8"), ("wInt", "<u8"), ("wIntLanes", "<u8"), ("wLeaf", "<u8"), ("wLeafSvc", "<u8")]))
a = a[a["t1"] > 0]
dur = (a["t1"] - a["t0"]).astype(np.float64) / 100.0  # wall_clock64 ticks at 100 MHz -> microseconds
span = (a["t1"].max() - a["t0"].min()) / 100.0
print(f"waves {len(a)}, kernel span {span:.1f} us")
print("wave lifetime us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
mx, sm = a["mx"].astype(np.float64), a["sm"].astype(np.float64)
heavy = mx > 8
print(f"heavy waves (>8 iterations): {heavy.sum()}, max-iter mean {mx[heavy].mean():.1f} p99 {np.percentile(mx[heavy], 99):.0f} max {mx.max():.0f}")
print(f"lane utilisation in heavy waves: {(sm[heavy].sum() / (64 * mx[heavy].sum())):.3f}")
print(f"us per iteration in heavy waves: {(dur[heavy] / mx[heavy]).mean():.3f}")
print(f"sum of wave lifetimes {dur.sum() / 1e3:.1f} ms -> mean resident waves {dur.sum() / span:.0f}")
# when do heavy waves start/end relative to the kernel
t0 = (a["t0"] - a["t0"].min()) / 100.0
for lo in range(0, int(span) + 1, max(1, int(span / 12))):
    sel = (t0 <= lo) & ((a["t1"] - a["t0"].min()) / 100.0 > lo)
    print(f"t={lo:5d} us resident waves {sel.sum():5d} heavy {(sel & heavy).sum():5d}")

wInt, wIntL, wLeaf = a["wInt"].astype(np.float64), a["wIntLanes"].astype(np.float64), a["wLeaf"].astype(np.float64)
wLeafL, wSvc = (a["wLeafSvc"] & 0xffffffff).astype(np.float64), (a["wLeafSvc"] >> 32).astype(np.float64)
h = heavy
print(f"wave-level steps in heavy waves: internal {wInt[h].sum():.0f} (active lanes/step {wIntL[h].sum() / wInt[h].sum():.1f}), "
      f"leaf {wLeaf[h].sum():.0f} (active lanes/step {wLeafL[h].sum() / max(1, wLeaf[h].sum()):.1f}), service {wSvc[h].sum():.0f}")
print(f"per heavy wave: internal steps {wInt[h].mean():.1f}, leaf steps {wLeaf[h].mean():.1f}, service {wSvc[h].mean():.1f}, max lane iters {mx[h].mean():.1f}")
print(f"us per wave-level step (internal+leaf): {(dur[h].sum() / (wInt[h].sum() + wLeaf[h].sum())):.3f}")
# the waves that end last: what their time is made of
end = (a["t1"] - a["t0"].min()) / 100.0
for i in np.argsort(-end)[:12]:
    print(f"ends {end[i]:6.1f} us  starts {t0[i]:6.1f}  lives {dur[i]:6.1f}  internal {wInt[i]:4.0f} @ {wIntL[i] / max(1, wInt[i]):4.1f} lanes  leaf {wLeaf[i]:3.0f} @ {wLeafL[i] / max(1, wLeaf[i]):4.1f}  service {wSvc[i]:3.0f}  "
          f"us/step {dur[i] / max(1, wInt[i] + wLeaf[i] + wSvc[i]):.3f}  longest lane {mx[i]:.0f} iterations")
steps = wInt + wLeaf + wSvc
print("steps of a wave: p50 %.0f p90 %.0f p99 %.0f max %.0f" % (*np.percentile(steps[h], [50, 90, 99]), steps.max()))
