#!/usr/bin/env python3
"""Writes tests/golden/s0_small_64x64.npz: outputs of the CPU oracle for frames 0..3 of scene S0 (Suzanne light +
8.7k-triangle dragon) at 64x64 with the committed Unity scene's settings and FrameRand = (12345 + 7919*k) mod
999999.  The reference itself holds no expected outputs (it has no tests and cannot run here), so these vectors
pin the oracle against regressions and let the GPU tests run without executing the oracle.

Also writes tests/golden/ris_small_64x48.npz: the same for the RIS project's kernels (GenerateCandidates, TraceRays,
AccumulateFrames), frames 0..2, 4 triangle candidates per pixel and 3 reservoir tests.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as h  # noqa: E402

scene = h.rt.build_scene_s0(subdivide=False)
cfg = h.small_config(64, 64)
orc = h.OracleRenderer(scene.arrays(), cfg, threads=1)
out = {}
for k in range(4):
    orc.render_frame()
    for name in ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer"):
        out[f"f{k}_{name}"] = orc.buf[name].view(np.uint8).copy()
    for name in ("Result", "AccumulatedResult", "DepthBuffer", "DebugOverlay", "NormalsBuffer"):
        out[f"f{k}_{name}"] = orc.tex[name].copy()
    out[f"f{k}_records"] = orc.records.view(np.uint8).copy()
dst = os.path.join(ROOT, "tests", "golden", "s0_small_64x64.npz")
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst), "bytes")

cfg = h.small_ris_config(64, 48, tri_candidates=4, tests=3)
orc = h.RisOracleRenderer(scene.arrays(), cfg, threads=1)
out = {}
for k in range(3):
    orc.render_frame()
    for name in ("HitInfos", "Candidates"):
        out[f"f{k}_{name}"] = orc.buf[name].view(np.uint8).copy()
    for name in ("Result", "AccumulatedResult", "DebugOverlay", "RayHitPoints"):
        out[f"f{k}_{name}"] = orc.tex[name].copy()
    out[f"f{k}_records"] = orc.records.view(np.uint8).copy()
out["world_normals"] = orc.tris["worldNormal"].copy()
dst = os.path.join(ROOT, "tests", "golden", "ris_small_64x48.npz")
np.savez_compressed(dst, **out)
print(dst, os.path.getsize(dst), "bytes")
