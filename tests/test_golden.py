"""Committed golden vectors (tests/golden/s0_small_64x64.npz and ris_small_64x48.npz, written by tools/make_golden.py): the oracle must keep
reproducing them bit for bit (CPU test), and the HIP path must match them without the oracle in the loop (GPU
test)."""
import os

import numpy as np
import pytest

import harness as h

rt = h.rt
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s0_small_64x64.npz")
BUFS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXS = ("Result", "AccumulatedResult", "DepthBuffer", "DebugOverlay", "NormalsBuffer")


def same(a, b):
    return np.array_equal(h.bits(a), h.bits(b))


def test_oracle_reproduces_golden(scene_small):
    g = np.load(GOLDEN)
    orc = h.OracleRenderer(scene_small.arrays(), h.small_config(64, 64), threads=0)   # all cores: also checks that
    for k in range(4):                                                                 # threading does not change bits
        orc.render_frame()
        for name in BUFS:
            assert same(orc.buf[name], g[f"f{k}_{name}"]), (k, name)
        for name in TEXS:
            assert same(orc.tex[name], g[f"f{k}_{name}"]), (k, name)
        assert same(orc.records, g[f"f{k}_records"]), (k, "records")


def test_golden_is_not_vacuous():
    g = np.load(GOLDEN)
    depth = g["f3_DepthBuffer"]
    rec = g["f3_records"].view(rt.DTYPES["TraceRecord"])
    assert np.isfinite(depth).sum() > 500
    assert len(np.unique(rec["triIndex"])) > 300 and (rec["lightTri"] >= 0).sum() > 500
    res = g["f3_Result"][..., :3]
    assert np.nanmax(res) > 0


@pytest.mark.gpu
def test_gpu_matches_golden(gpu_ctx, scene_small):
    g = np.load(GOLDEN)
    cfg = h.small_config(64, 64)
    drv = rt.Driver(gpu_ctx, cfg, scene_small)
    rec = gpu_ctx.create_buffer(64 * 64, 32)
    for k in (0, 4):
        gpu_ctx.check(gpu_ctx.lib.rt_set_buffer(drv.shader("rt"), k, b"TraceRecords", rec), "bind TraceRecords")
    try:
        for k in range(4):
            drv.render_frame()
            gpu_ctx.sync()
            for name in BUFS:
                assert same(drv.buffer(name), g[f"f{k}_{name}"]), (k, name)
            for name in TEXS:
                assert same(drv.texture(name), g[f"f{k}_{name}"]), (k, name)
            assert same(gpu_ctx.get_buffer(rec, rt.DTYPES["TraceRecord"]), g[f"f{k}_records"]), (k, "records")
    finally:
        drv.close()
        gpu_ctx.lib.rt_buffer_release(rec)


# ---- RIS project ----
RIS_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ris_small_64x48.npz")
RIS_BUFS = ("HitInfos", "Candidates")
RIS_TEXS = ("Result", "AccumulatedResult", "DebugOverlay", "RayHitPoints")


def _ris_cfg():
    return h.small_ris_config(64, 48, tri_candidates=4, tests=3)


def test_ris_oracle_reproduces_golden(scene_small):
    g = np.load(RIS_GOLDEN)
    orc = h.RisOracleRenderer(scene_small.arrays(), _ris_cfg(), threads=0)
    assert same(orc.tris["worldNormal"], g["world_normals"])
    for k in range(3):
        orc.render_frame()
        for name in RIS_BUFS:
            assert same(orc.buf[name], g[f"f{k}_{name}"]), (k, name)
        for name in RIS_TEXS:
            assert same(orc.tex[name], g[f"f{k}_{name}"]), (k, name)
        assert same(orc.records, g[f"f{k}_records"]), (k, "records")
    assert np.nanmax(g["f2_Result"][..., :3]) > 0 and (g["f2_Candidates"].view(rt.DTYPES["Float4"])["w"] > 0).sum() > 500


@pytest.mark.gpu
def test_ris_gpu_matches_golden(gpu_ctx, scene_small):
    g = np.load(RIS_GOLDEN)
    drv = rt.RisDriver(gpu_ctx, _ris_cfg(), scene_small)
    rec = gpu_ctx.create_buffer(64 * 48, 32)
    for k in (0, 1):
        gpu_ctx.check(gpu_ctx.lib.rt_set_buffer(drv.shader("rt"), k, b"TraceRecords", rec), "bind TraceRecords")
    try:
        assert same(drv.buffer("Tris")["worldNormal"], g["world_normals"])
        for k in range(3):
            drv.render_frame()
            gpu_ctx.sync()
            for name in RIS_BUFS:
                assert same(drv.buffer(name), g[f"f{k}_{name}"]), (k, name)
            for name in RIS_TEXS:
                assert same(drv.texture(name), g[f"f{k}_{name}"]), (k, name)
            assert same(gpu_ctx.get_buffer(rec, rt.DTYPES["TraceRecord"]), g[f"f{k}_records"]), (k, "records")
    finally:
        drv.close()
        gpu_ctx.lib.rt_buffer_release(rec)
