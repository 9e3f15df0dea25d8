"""GPU parity of the RIS project's kernels (RIS_BVH_RayTracer: GenerateCandidates, TraceRays; its PreCalc and
PostProcesser) against the CPU oracle, through the C-ABI.  Everything is compared bit for bit."""
import numpy as np
import pytest

import harness as h

pytestmark = pytest.mark.gpu
rt = h.rt


def _compare_frame(drv, orc, what=""):
    for name in ("HitInfos", "Candidates"):
        assert np.array_equal(h.bits(drv.buffer(name)), h.bits(orc.buf[name])), (what, name)
    for name in ("Result", "DebugOverlay", "RayHitPoints", "AccumulatedResult"):
        assert np.array_equal(h.bits(drv.texture(name)), h.bits(orc.tex[name])), (what, name)


@pytest.mark.parametrize("res,tri_cand,obj_cand,tests,one_wave", [((64, 64), 4, 1, 3, None), ((72, 40), 1, 1, 1, None), ((40, 56), 3, 2, 0, None),
                                                                  ((72, 56), 9, 2, 2, 0), ((72, 56), 9, 2, 2, 31)])
def test_frames_match_oracle(gpu_ctx, scene_small, res, tri_cand, obj_cand, tests, one_wave):
    """`one_wave`: every kernel with four waves per workgroup (0) or one (31) instead of the default mix."""
    cfg = h.small_ris_config(*res, tri_candidates=tri_cand, tests=tests)
    cfg.SceneObjectCandidatesNum = obj_cand
    if one_wave is not None:
        gpu_ctx.set_tuning("one_wave_groups", one_wave)
    drv = rt.RisDriver(gpu_ctx, cfg, scene_small)
    try:
        orc = h.RisOracleRenderer(scene_small.arrays(), cfg)
        ris = drv.ris_arrays()
        assert np.array_equal(h.bits(ris["Materials"]), h.bits(orc.mats))
        # CalcTriNormals ran on upload: world normals must equal the oracle's
        assert np.array_equal(h.bits(drv.buffer("Tris")), h.bits(orc.tris))
        for frame in range(3):
            drv.render_frame()
            orc.render_frame()
            gpu_ctx.sync()
            _compare_frame(drv, orc, frame)
        hit = np.isfinite(orc.buf["HitInfos"]["dst"])
        assert hit.any()
        if tests > 0:   # with zero reservoir tests the correction weight is infinite and every pixel is black
            assert (orc.tex["Result"][..., :3] > 0).any()
    finally:
        drv.close()
        gpu_ctx.set_tuning("one_wave_groups", 15)


def test_trace_records_and_counters(gpu_ctx, scene_small):
    cfg = h.small_ris_config(64, 48, tri_candidates=2, tests=2)
    drv = rt.RisDriver(gpu_ctx, cfg, scene_small)
    rec = gpu_ctx.create_buffer(64 * 48, 32)
    try:
        for k in (0, 1):
            gpu_ctx.check(gpu_ctx.lib.rt_set_buffer(drv.shader("rt"), k, b"TraceRecords", rec), "bind records")
        gpu_ctx.enable_stats(2)
        orc = h.RisOracleRenderer(scene_small.arrays(), cfg)
        for _ in range(2):
            drv.render_frame()
            orc.render_frame()
        gpu_ctx.sync()
        got = gpu_ctx.get_buffer(rec, h.DT["TraceRecord"])
        for field in ("triIndex", "objIndex", "visitHash", "bounceVisitHash"):
            assert np.array_equal(got[field], orc.records[field]), field
        s0, s1 = gpu_ctx.get_stats(drv.shader("rt"), 0), gpu_ctx.get_stats(drv.shader("rt"), 1)
        c = orc.take_counters()
        for key in ("rays", "pops", "bvChecks", "objVisits", "triIters", "triChecks"):
            assert s0[key] + s1[key] == c[key], key
    finally:
        gpu_ctx.enable_stats(0)
        gpu_ctx.lib.rt_buffer_release(rec)
        drv.close()


def test_two_bounces_and_smoothness(gpu_ctx, scene_small):
    cfg = h.small_ris_config(48, 48, tri_candidates=2, tests=2)
    cfg.MaxBounceCount = 2
    cfg.smoothness = 0.6
    cfg.ScatterProbability = 0.5
    drv = rt.RisDriver(gpu_ctx, cfg, scene_small)
    try:
        orc = h.RisOracleRenderer(scene_small.arrays(), cfg)
        for frame in range(2):
            drv.render_frame()
            orc.render_frame()
            gpu_ctx.sync()
            _compare_frame(drv, orc, frame)
    finally:
        drv.close()


def test_row_bands_match_full_frame(gpu_ctx, scene_small):
    cfg = h.small_ris_config(64, 64, tri_candidates=2, tests=1)
    full = rt.RisDriver(gpu_ctx, cfg, scene_small)
    full.render_frame()
    ref = {n: full.texture(n) for n in ("Result", "RayHitPoints")}
    ref_c = full.buffer("Candidates")
    full.close()
    band = rt.RisDriver(gpu_ctx, cfg, scene_small, 16, 40)
    try:
        band.render_frame()
        for n in ref:
            got = band.texture(n)
            assert np.array_equal(h.bits(got[16:40]), h.bits(ref[n][16:40])), n
            assert (got[:16] == 0).all() and (got[40:] == 0).all()
        lo, hi = 16 * 64 * 2, 40 * 64 * 2
        assert np.array_equal(h.bits(band.buffer("Candidates")[lo:hi]), h.bits(ref_c[lo:hi]))
    finally:
        gpu_ctx.set_row_band(0, 0)
        band.close()


def test_unbound_and_bad_strides_are_reported(gpu_ctx):
    sh = gpu_ctx.open_shader("RIS_BVH_RayTracer")
    lib = gpu_ctx.lib
    try:
        assert lib.rt_find_kernel(sh, b"GenerateCandidates") == 0 and lib.rt_find_kernel(sh, b"TraceRays") == 1
        assert lib.rt_find_kernel(sh, b"InitialTrace") < 0
        res = (h.C.c_int * 2)(16, 16)
        lib.rt_set_ints(sh, b"Resolution", res, 2)
        assert lib.rt_dispatch(sh, 0, 2, 2, 1) == -4            # RT_ERR_UNBOUND
        assert b"needs BVs, Tris" in lib.rt_last_error(gpu_ctx.h)
        assert lib.rt_dispatch(sh, 2, 2, 2, 1) == -1            # no such kernel
    finally:
        lib.rt_shader_close(sh)
