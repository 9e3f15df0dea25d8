"""More seeds for the randomised parity tests (tests/test_gpu_fuzz.py) than the suite runs, with the measured
alternatives switched on for some of them (top nodes in LDS, SpatialReusePass staged in LDS, three frames in flight).

    python tools/fuzz_more.py [first_seed count]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as h  # noqa: E402
import test_gpu_fuzz as fz  # noqa: E402

first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 60)
for seed in range(first, first + count):
    ctx = h.rt.RtContext(0)
    try:
        note = ""
        if seed % 3 == 0:
            ctx.set_tuning("top_nodes_lds", (8, 33, 64)[seed // 3 % 3]); note += " top_nodes_lds"
        if seed % 4 == 1:
            ctx.set_tuning("spatial_lds_max_offset", 3); note += " spatial_lds"
        if seed % 5 == 2:
            ctx.set_tuning("frames_in_flight", 3); note += " frames_in_flight=3"
        fz.test_random_scenes_restir(ctx, seed)
        fz.test_random_scenes_ris(ctx, seed)
        print("seed", seed, "ok" + note, flush=True)
    finally:
        ctx.close()
print("ok")
