"""GPU parity on the benchmark configurations themselves (the five `configs` of the baseline description).

  1. ReSTIR project, 80kScene* stand-in, 256x256, TriReservoirTestsNum = 32: the whole frame against the whole oracle.
  2. RIS project, 80kScene*, 1920x1080, 32 candidates per pixel, 32 reservoir tests: rows sampled over the image.
  3. ReSTIR 1080p with temporal + spatial reuse: tests/test_gpu_fullsize.py.
  4. ReSTIR 3840x2160 in row bands: tests/test_gpu_fullsize.py (two bands) and the multi-rank schedule tests.
  5. 128 dragons (10 M triangles through one shared mesh BVH, or 128 distinct meshes) + 10 x 1000 emissive triangles at
     3840x2160: determinism, band invariance, and oracle parity on sampled rows.
At the sizes where the whole oracle would take minutes, the oracle runs only on a few rows: before each stage its
inputs for those rows (and the rows around them that SpatialReusePass reads) are overwritten with the GPU's, both run
the stage, and the rows must agree bit for bit.  Every kernel is per pixel, so a row of a full-size frame is a full-size
result.
"""
import numpy as np
import pytest

import harness as h
from test_gpu_parity import attach_records, compare_all

rt = h.rt
pytestmark = pytest.mark.gpu

BUFS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXS = ("DepthBuffer", "NormalsBuffer", "DebugOverlay", "RayHitPointsA", "RayHitPointsB", "Result", "AccumulatedResult")


# ----------------------------------------------------------------------------------------------------- config 1
@pytest.mark.parametrize("reuse", [True, False])
def test_config1_256x256_32_candidates(gpu_ctx, scene_80k, reuse):
    cfg = h.small_config(256, 256)
    cfg.TriReservoirTestsNum = 32
    cfg.SceneObjectReservoirTestsNum = 5
    if not reuse:   # "1 spp primary hit + RIS DL": InitialTrace and TraceRays alone
        cfg.TemporalReuseWeight = 0.0
        cfg.SpatialReuseIterations = 0
    drv = rt.Driver(gpu_ctx, cfg, scene_80k)
    orc = h.OracleRenderer(scene_80k.arrays(), cfg)
    rec = attach_records(gpu_ctx, drv, 256 * 256)
    try:
        for frame in range(4 if reuse else 2):
            drv.render_frame()
            orc.render_frame()
            compare_all(gpu_ctx, drv, orc, rec, what=f"config 1 reuse={reuse} frame {frame}")
        hit = np.isfinite(orc.tex["DepthBuffer"])
        assert hit.sum() > 0.1 * hit.size
        assert (orc.records["lightTri"] >= 0).sum() == hit.sum()           # every hit pixel drew its 32 + 1 triangles
        assert len(np.unique(orc.records["lightTri"][orc.records["lightTri"] >= 0])) > 300
    finally:
        drv.close()
        gpu_ctx.lib.rt_buffer_release(rec)


# ----------------------------------------------------------------------------------------------------- config 2
RIS_ROWS = [(0, 8), (300, 308), (536, 552), (800, 808), (1072, 1080)]


def test_config2_ris_1080p_32_candidates(gpu_ctx, scene_80k):
    cfg = rt.ris_default_config()
    cfg.Resolution[0], cfg.Resolution[1] = 1920, 1080
    cfg.TriCandidatesNum = 32
    cfg.CandidateReservoirTestsNum = 32
    W, tot = 1920, 32
    drv = rt.RisDriver(gpu_ctx, cfg, scene_80k)
    orc = h.RisOracleRenderer(scene_80k.arrays(), cfg, records=False)
    try:
        for frame in range(2):
            drv.update_per_frame()
            orc.update_per_frame()
            drv.run_stage(0)
            drv.run_stage(1)
            drv.run_post_processing()
            gpu_ctx.sync()
            hits = 0
            for r0, r1 in RIS_ROWS:   # the RIS kernels read only their own pixel: the oracle runs the rows on its own
                orc.set_band(r0, r1)
                orc.generate_candidates()
                orc.trace_rays()
                orc.accumulate()
                px = slice(r0 * W, r1 * W)
                got = gpu_ctx.get_buffer_range(drv.buffer_handle("HitInfos"), h.DT["RisHitInfo"], r0 * W, (r1 - r0) * W)
                assert np.array_equal(h.bits(got), h.bits(orc.buf["HitInfos"][px])), (frame, r0, "HitInfos")
                got = gpu_ctx.get_buffer_range(drv.buffer_handle("Candidates"), h.DT["Float4"], r0 * W * tot, (r1 - r0) * W * tot)
                assert np.array_equal(h.bits(got), h.bits(orc.buf["Candidates"][r0 * W * tot:r1 * W * tot])), (frame, r0, "Candidates")
                for name in ("Result", "AccumulatedResult", "DebugOverlay", "RayHitPoints"):
                    got = gpu_ctx.get_texture_rows(drv.texture_handle(name), W, r0, r1, 4)
                    assert np.array_equal(h.bits(got), h.bits(orc.tex[name][r0:r1])), (frame, r0, name)
                hits += int(np.isfinite(orc.buf["HitInfos"]["dst"][px]).sum())
            assert hits > 2000, "the sampled rows miss the scene"
    finally:
        drv.close()


# ----------------------------------------------------------------------------------------------------- config 5
@pytest.fixture(scope="module")
def scene_instanced():
    return rt.build_scene_instanced(grid=(16, 8), emissive_objects=10, emissive_tris=1000, unique=False, subdivide=True)


def partial_snapshot(drv, bufs, texs):
    drv.ctx.sync()
    return {**{n: drv.buffer(n) for n in bufs}, **{n: drv.texture(n) for n in texs}}


def banded_frame(ctx, drv, cfg, bands):
    """One frame rendered band by band, dispatch by dispatch: the order a set of GPUs would produce."""
    drv.update_per_frame()
    steps = ([lambda: drv.run_stage(0)], [lambda: drv.run_stage(1)],
             *[[lambda i=i: drv.run_spatial_pass(i)] for i in range(cfg.SpatialReuseIterations)],
             [drv.run_transfer], [lambda: drv.run_stage(3), drv.run_post_processing])
    for step in steps:
        for b in bands:
            ctx.set_row_band(*b)
            for f in step:
                f()
    ctx.set_row_band(0, 0)


def determinism_and_bands(ctx, scene, frames=2):
    cfg = rt.default_config()
    cfg.Resolution[0], cfg.Resolution[1] = 3840, 2160
    bufs, texs = ("Candidates", "HitInfos"), ("Result", "DepthBuffer")
    runs = []
    for _ in range(2):
        drv = rt.Driver(ctx, cfg, scene)
        try:
            for _ in range(frames):
                drv.render_frame()
            runs.append(partial_snapshot(drv, bufs, texs))
        finally:
            drv.close()
    for k in runs[0]:
        assert np.array_equal(h.bits(runs[0][k]), h.bits(runs[1][k])), f"second run: {k} differs"
    depth = runs[0]["DepthBuffer"]
    assert 0.2 * depth.size < np.isfinite(depth).sum() < depth.size
    drv = rt.Driver(ctx, cfg, scene)
    try:
        for _ in range(frames):
            banded_frame(ctx, drv, cfg, [(0, 712), (712, 1448), (1448, 2160)])
        got = partial_snapshot(drv, bufs, texs)
    finally:
        ctx.set_row_band(0, 0)
        drv.close()
    for k in got:
        assert np.array_equal(h.bits(runs[0][k]), h.bits(got[k])), f"three bands: {k} differs"
    return runs[0]


def test_config5_instanced_4k_determinism_and_bands(gpu_ctx, scene_instanced):
    a = scene_instanced.arrays()
    assert len(a["SceneObjects"]) == 138 and len(a["LightObjects"]) == 10
    assert sum(int(a["ComponentDatas"][o["bvStartIndex"]][1]) for o in a["SceneObjects"]) > 10_000_000
    determinism_and_bands(gpu_ctx, scene_instanced)


def test_config5_unique_4k_determinism_and_bands(gpu_ctx):
    """128 distinct (vertex-jittered) dragon meshes, each with its own BVH: 10 M stored triangles."""
    scene = rt.build_scene_instanced(grid=(16, 8), emissive_objects=10, emissive_tris=1000, unique=True, subdivide=True)
    assert scene._data.nTriangles > 10_000_000
    determinism_and_bands(gpu_ctx, scene)


HALO = 29  # SpatialReusePass reads up to MaxOffset (27 in the last pass) rows away, one more through the x wrap


@pytest.mark.parametrize("rows", [(0, 8), (400, 408), (1080, 1096), (2152, 2160)])
def test_config5_instanced_4k_oracle_rows(gpu_ctx, scene_instanced, rows):
    """Frames 0..1 of the 10 M-triangle scene at 3840x2160, stage by stage, oracle on the sampled rows only; the rows it
    reads (the sample and HALO rows around it) are refreshed from the GPU before every stage."""
    cfg = rt.default_config()
    cfg.Resolution[0], cfg.Resolution[1] = 3840, 2160
    W, H = 3840, 2160
    drv = rt.Driver(gpu_ctx, cfg, scene_instanced)
    orc = h.OracleRenderer(scene_instanced.arrays(), cfg, records=False, row0=rows[0], row1=rows[1])
    lo, hi = max(0, rows[0] - HALO), min(H, rows[1] + HALO)

    def load_gpu_state():
        gpu_ctx.sync()
        for n in BUFS:
            orc.buf[n][lo * W:hi * W] = drv.buffer_rows(n, lo, hi)
        for n in TEXS:
            orc.tex[n][lo:hi] = drv.texture_rows(n, lo, hi)

    def compare(what):
        gpu_ctx.sync()
        for n in BUFS:
            assert np.array_equal(h.bits(drv.buffer_rows(n, *rows)), h.bits(orc.buf[n][rows[0] * W:rows[1] * W])), f"{what}: {n}"
        for n in TEXS:
            assert np.array_equal(h.bits(drv.texture_rows(n, *rows)), h.bits(orc.tex[n][rows[0]:rows[1]])), f"{what}: {n}"

    try:
        for frame in range(2):
            drv.update_per_frame()
            orc.update_per_frame()
            for stage in (0, 1):
                load_gpu_state()
                drv.run_stage(stage)
                orc.run_stage(stage)
                compare(f"frame {frame} stage {stage}")
            cycle = False
            for i in range(cfg.SpatialReuseIterations):
                cycle = not cycle
                load_gpu_state()
                drv.run_spatial_pass(i)
                orc.spatial_pass(i, cycle)
                compare(f"frame {frame} spatial pass {i}")
            load_gpu_state()
            drv.run_transfer()
            orc.transfer_to_original()
            compare(f"frame {frame} transfer")
            load_gpu_state()
            drv.run_stage(3)
            orc.trace_rays()
            compare(f"frame {frame} trace rays")
            load_gpu_state()
            drv.run_post_processing()
            orc.accumulate()
            compare(f"frame {frame} accumulate")
        if rows[0] < 2000:   # the scene fills the image rows below about 1100; the last sample is sky
            assert np.isfinite(orc.tex["DepthBuffer"][rows[0]:rows[1]]).sum() > 1000, "the sampled rows miss the scene"
    finally:
        drv.close()
