"""Randomised parity: seeded random scenes (triangle soups with duplicate, degenerate and axis-aligned flat
triangles; random rotations and non-uniform scales; several lights; an axis-aligned camera whose central ray has
zero direction components) rendered by the HIP path through the C-ABI and by the CPU oracle.  Every buffer and
render target of both projects must agree bit for bit, NaNs included."""
import numpy as np
import pytest

import harness as h

rt = h.rt
pytestmark = pytest.mark.gpu

BUFS = ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer")
TEXS = ("DepthBuffer", "NormalsBuffer", "DebugOverlay", "RayHitPointsA", "RayHitPointsB", "Result", "AccumulatedResult")


def random_mesh(rng, ntris, flavour):
    """Triangle soup in [-1, 1]^3.  flavour 0: small random triangles; 1: adds exact duplicates and zero-area
    triangles; 2: axis-aligned flat quads (boxes of zero thickness) mixed in; 3: a few huge overlapping triangles."""
    centres = rng.uniform(-1, 1, (ntris, 1, 3))
    size = 0.25 if flavour != 3 else 1.5
    tri = (centres + rng.uniform(-size, size, (ntris, 3, 3))).astype(np.float32)
    if flavour == 1:
        tri[ntris // 2:ntris // 2 + ntris // 8] = tri[:ntris // 8]                 # exact duplicates: distance ties
        tri[-ntris // 10:, 2] = tri[-ntris // 10:, 1]                                 # two equal corners: zero area
    if flavour == 2:
        k = ntris // 3
        axis = rng.integers(0, 3, k)
        for i in range(k):
            tri[i, :, axis[i]] = np.float32(round(float(tri[i, 0, axis[i]]) * 4) / 4)   # flat on a grid plane
    verts = np.zeros(ntris * 3, rt.DTYPES["Vertex"])
    verts["pos"] = tri.reshape(-1, 3)
    verts["uv"] = rng.uniform(0, 1, (ntris * 3, 2)).astype(np.float32)
    idx = np.arange(ntris * 3, dtype=np.int32).reshape(-1, 3)
    return rt.Mesh.from_arrays(verts, idx)


def random_scene(seed):
    rng = np.random.default_rng(seed)
    s = rt.Scene(split_resolution=int(rng.integers(2, 7)), max_depth_scene_bvh=int(rng.integers(1, 6)))
    mats = [s.add_material(0.0, rng.uniform(0.1, 1, 3), roughness=float(rng.uniform(0, 1)), metallicity=float(rng.uniform(0, 1)))
            for _ in range(2)]
    lights = [s.add_material(float(rng.uniform(1e3, 5e4)), rng.uniform(0.5, 1, 3)) for _ in range(2)]
    nobj = int(rng.integers(3, 8))
    for o in range(nobj):
        mesh = s.add_mesh(random_mesh(rng, int(rng.integers(12, 300)), int(rng.integers(0, 4))))
        emissive = o < 2
        pos = rng.uniform(-4, 4, 3) + np.array([0, 0, 10.0])
        scale = rng.uniform(0.5, 3.0, 3) if o % 2 else np.full(3, float(rng.uniform(0.5, 3.0)))
        s.add_object(mesh, pos, rng.uniform(-180, 180, 3), scale, (lights if emissive else mats)[o % 2],
                     int(rng.integers(1, 20)))
    return s.build(threads=2)


def axis_camera(cfg):
    cfg.cameraPosition[0], cfg.cameraPosition[1], cfg.cameraPosition[2] = 0.0, 0.0, 0.0
    cfg.cameraRotation[0], cfg.cameraRotation[1], cfg.cameraRotation[2], cfg.cameraRotation[3] = 0.0, 0.0, 0.0, 1.0


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_scenes_restir(gpu_ctx, seed):
    scene = random_scene(seed)
    cfg = h.small_config(66, 50)     # even sizes: pixel (33, 25) looks exactly down +z
    axis_camera(cfg)
    cfg.MaxBounceCount = 1 + seed % 2
    drv = rt.Driver(gpu_ctx, cfg, scene)
    orc = h.OracleRenderer(scene.arrays(), cfg)
    rec = gpu_ctx.create_buffer(66 * 50, 32)
    try:
        for k in (0, 4):
            gpu_ctx.check(gpu_ctx.lib.rt_set_buffer(drv.shader("rt"), k, b"TraceRecords", rec), "bind TraceRecords")
        for frame in range(3):
            drv.render_frame()
            orc.render_frame()
            gpu_ctx.sync()
            for n in BUFS:
                assert np.array_equal(h.bits(drv.buffer(n)), h.bits(orc.buf[n])), (seed, frame, n)
            for n in TEXS:
                assert np.array_equal(h.bits(drv.texture(n)), h.bits(orc.tex[n])), (seed, frame, n)
            assert np.array_equal(h.bits(gpu_ctx.get_buffer(rec, rt.DTYPES["TraceRecord"])), h.bits(orc.records)), (seed, frame)
        assert np.isfinite(orc.tex["DepthBuffer"]).any(), "the random scene is not in view"
    finally:
        gpu_ctx.lib.rt_buffer_release(rec)
        drv.close()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_scenes_ris(gpu_ctx, seed):
    scene = random_scene(seed)
    cfg = h.small_ris_config(66, 50, tri_candidates=3, tests=2)
    cfg.SceneObjectCandidatesNum = 2
    cfg.cameraPosition[0], cfg.cameraPosition[1], cfg.cameraPosition[2] = 0.0, 0.0, 0.0
    cfg.cameraEulerDeg[0], cfg.cameraEulerDeg[1], cfg.cameraEulerDeg[2] = 0.0, 0.0, 0.0
    drv = rt.RisDriver(gpu_ctx, cfg, scene)
    orc = h.RisOracleRenderer(scene.arrays(), cfg, records=False)
    try:
        assert np.array_equal(h.bits(drv.buffer("Tris")), h.bits(orc.tris))
        for frame in range(2):
            drv.render_frame()
            orc.render_frame()
            gpu_ctx.sync()
            for n in ("HitInfos", "Candidates"):
                assert np.array_equal(h.bits(drv.buffer(n)), h.bits(orc.buf[n])), (seed, frame, n)
            for n in ("Result", "DebugOverlay", "RayHitPoints", "AccumulatedResult"):
                assert np.array_equal(h.bits(drv.texture(n)), h.bits(orc.tex[n])), (seed, frame, n)
        assert np.isfinite(orc.buf["HitInfos"]["dst"]).any(), "the random scene is not in view"
    finally:
        drv.close()
