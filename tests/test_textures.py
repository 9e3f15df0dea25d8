"""Texture atlas construction (ObjectHelper.ConstructTextureAtlas, ObjectHelper.cs:370-452) and the kernels'
atlas-sampling paths: colour, bump, roughness, metallicity and normal-map textures."""
import numpy as np
import pytest

import harness as h

rt = h.rt


@pytest.fixture(scope="module")
def scene_textured():
    return rt.build_scene_textured()


def test_atlas_holds_every_texture_where_the_material_says(scene_textured):
    a = scene_textured.arrays()
    atlas, mats = a["atlas"], a["Materials"]
    assert atlas.shape[0] == atlas.shape[1] and atlas.shape[0] & (atlas.shape[0] - 1) == 0      # power-of-two square
    tex = rt.procedural_textures()
    m = mats[1]
    assert tuple(m["col"]) == (-1, -1, -1) and m["bump"] == -1 and m["roughness"] == -1 and m["metallicity"] == 1.0
    for slot, loc, dims in (("col", m["colTexLoc"], m["colTexDims"]), ("bump", m["bumpTexLoc"], m["bumpTexDims"]),
                            ("roughness", m["roughnessTexLoc"], m["roughnessTexDims"]),
                            ("normals", m["normalsTexLoc"], m["normalsTexDims"])):
        t = tex[slot]
        assert tuple(dims) == (t.shape[1], t.shape[0]), slot
        assert np.array_equal(atlas[loc[1]:loc[1] + dims[1], loc[0]:loc[0] + dims[0]], t), slot
    assert tuple(mats[0]["normalsTexLoc"]) == (-1, -1) and tuple(mats[0]["col"]) == (1, 1, 0)   # untextured: scalars
    assert tuple(mats[2]["col"]) == (-1, -1, -1) and tuple(mats[2]["colTexDims"]) == (24, 16)
    # rectangles do not overlap (1 texel of padding between them)
    used = np.zeros(atlas.shape[:2], np.int32)
    for mm in mats:
        for key in ("col", "bump", "roughness", "normals"):
            loc, dims = mm[key + "TexLoc"], mm[key + "TexDims"]
            if dims[0] > 0 and loc[0] >= 0:
                used[loc[1]:loc[1] + dims[1] + 1, loc[0]:loc[0] + dims[0] + 1] += 1
    assert used.max() == 1


def test_metallicity_texture_cannot_be_built():
    """The reference never packs metallicity textures but takes a rectangle for them (ObjectHelper.cs:373-379 against
    :430-437), so its atlas builder runs out of rectangles: the C# throws, rth_scene_build fails."""
    s = rt.Scene()
    t = rt.procedural_textures()
    s.add_material(1.0, (1, 1, 1), textures={"col": t["col"], "metallicity": t["roughness"]})
    m = s.add_mesh(rt.Mesh.disc(8))
    s.add_object(m, (0, 0, 5), (0, 0, 0), (1, 1, 1), 0, 4)
    with pytest.raises(rt.NativeError):
        s.build()


def test_oracle_uses_the_textures(scene_textured, scene_small):
    cfg = h.small_config(64, 64)
    tex, plain = h.OracleRenderer(scene_textured.arrays(), cfg), h.OracleRenderer(scene_small.arrays(), cfg)
    for _ in range(2):
        tex.render_frame()
        plain.render_frame()
    hit = np.isfinite(tex.tex["DepthBuffer"])
    assert hit.any()
    # normal map: shading normals differ from the geometric ones; bump: hit points move along the ray
    assert (tex.buf["HitInfos"]["normal"] != plain.buf["HitInfos"]["normal"]).any()
    assert (tex.buf["HitInfos"]["hitPoint"] != plain.buf["HitInfos"]["hitPoint"]).any()
    assert not np.array_equal(tex.tex["Result"], plain.tex["Result"])
    n = tex.buf["HitInfos"]["normal"][hit.reshape(-1)]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-4)


def _with_metallicity_texture(arrays):
    """Material 1 additionally reads its metallicity from the atlas (the roughness texture's rectangle): built by hand,
    since the reference's atlas builder cannot produce it."""
    a = dict(arrays)
    mats = a["Materials"].copy()
    mats[1]["metallicity"] = -1
    mats[1]["metallicityTexLoc"] = mats[1]["roughnessTexLoc"]
    mats[1]["metallicityTexDims"] = mats[1]["roughnessTexDims"]
    a["Materials"] = mats
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("metal_tex", [False, True])
def test_gpu_matches_oracle_with_textures(gpu_ctx, scene_textured, metal_tex):
    cfg = h.small_config(96, 64)
    arrays = scene_textured.arrays()
    drv = rt.Driver(gpu_ctx, cfg, scene_textured)
    try:
        if metal_tex:
            arrays = _with_metallicity_texture(arrays)
            gpu_ctx.set_buffer_data(drv.buffer_handle("Materials"), arrays["Materials"])
        orc = h.OracleRenderer(arrays, cfg)
        for frame in range(3):
            drv.render_frame()
            orc.render_frame()
            gpu_ctx.sync()
            for name in ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer"):
                assert np.array_equal(h.bits(drv.buffer(name)), h.bits(orc.buf[name])), (frame, name)
            for name in ("Result", "AccumulatedResult", "DepthBuffer", "NormalsBuffer", "DebugOverlay"):
                assert np.array_equal(h.bits(drv.texture(name)), h.bits(orc.tex[name])), (frame, name)
    finally:
        drv.close()
