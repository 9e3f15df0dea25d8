"""The JSON scene cache (FileLoader.cs:12-91): reading the reference's own files, write -> read round trip, errors.

tests/golden/80kScene_cache/ holds four small DATA files of the dump committed with the reference
(BVH_ReSTIR_DL/80kScene/80kScene_{lightObjects,sceneObjectDatas,integers,floats}.json), byte for byte; the large
arrays of that dump are either missing from the reference (loadedBVs, loadedVertices) or covered elsewhere
(tests/golden/ref_80kscene.json.gz)."""
import json
import os
import shutil

import numpy as np
import pytest

import harness as h

rt = h.rt
REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "80kScene_cache")
ARRAYS = ("BVs", "Vertices", "Triangles", "SceneObjects", "LightObjects", "Materials", "ComponentDatas")


def _empty_like(scene_small):
    """A scene with the same material inputs as S0 and nothing else."""
    s = rt.Scene()
    for m in scene_small.arrays()["Materials"]:
        s.add_material(float(m["brightness"]), [float(c) for c in m["col"]], float(m["roughness"]), float(m["metallicity"]), float(m["bump"]))
    return s


def test_round_trip_is_bit_exact(scene_small, tmp_path):
    scene_small.save_json(str(tmp_path))
    assert json.load(open(tmp_path / "renderBVs.json")) == {"array": []}     # as the reference's writer leaves it
    loaded = _empty_like(scene_small).load_json(str(tmp_path))
    a, b = scene_small.arrays(), loaded.arrays()
    for k in ARRAYS:
        assert np.array_equal(h.bits(a[k]), h.bits(b[k])), k
    for k in ("maxBVHDepth", "emittingObjectsNum", "sceneBVHStartIndex", "totArea"):
        assert a[k] == b[k], k


def test_reads_the_reference_dump(scene_small, tmp_path):
    scene_small.save_json(str(tmp_path))
    for k in ("lightObjects", "sceneObjectDatas", "integers", "floats"):
        os.remove(tmp_path / f"{k}.json")
        shutil.copy(os.path.join(REF, f"80kScene_{k}.json"), tmp_path)
    loaded = _empty_like(scene_small).load_json(str(tmp_path), prefix="80kScene")
    b = loaded.arrays()
    ref_objs = json.load(open(os.path.join(REF, "80kScene_sceneObjectDatas.json")))["array"]
    assert len(b["SceneObjects"]) == len(ref_objs) == 2
    for got, want in zip(b["SceneObjects"], ref_objs):
        for name, key in (("worldToLocal", "worldToLocalMatrix"), ("localToWorld", "localToWorldMatrix")):
            m = np.array([[want[key][f"c{c}"][r] for r in "xyzw"] for c in range(4)], np.float64).astype(np.float32).reshape(16)
            assert np.array_equal(got[name], m)                    # column-major, c0 first
        assert got["materialIndex"] == want["materialIndex"] and got["bvStartIndex"] == want["bvStartIndex"]
        assert got["areaApprox"] == np.float32(want["areaApprox"])
        assert np.array_equal(got["min"], np.array([want["min"][c] for c in "xyz"], np.float32))
    ref_light = json.load(open(os.path.join(REF, "80kScene_lightObjects.json")))["array"][0]
    assert b["LightObjects"][0]["triStart"] == ref_light["triStart"] and b["LightObjects"][0]["totTris"] == ref_light["totTris"] == 968
    assert b["LightObjects"][0]["brightness"] == np.float32(ref_light["brightness"])
    assert (b["maxBVHDepth"], b["emittingObjectsNum"], b["sceneBVHStartIndex"]) == (18, 1, 18522)
    assert b["totArea"] == float(np.float32(555.0943603515625))
    # the builder reproduces what the dump stores for the object it can be compared on exactly (the dragon: identity
    # rotation), including the area the reference's runtime rounds once from double precision
    a = scene_small.arrays()
    assert np.array_equal(h.bits(a["SceneObjects"][0]), h.bits(b["SceneObjects"][0]))
    assert a["totArea"] == b["totArea"] and a["sceneBVHStartIndex"] == b["sceneBVHStartIndex"]


def test_written_text_matches_the_reference_files(scene_small, tmp_path):
    scene_small.save_json(str(tmp_path))
    for k in ("integers", "floats"):      # same values, same layout: the text must be identical
        assert open(tmp_path / f"{k}.json").read() == open(os.path.join(REF, f"80kScene_{k}.json")).read(), k
    mine = open(tmp_path / "sceneObjectDatas.json").read().splitlines()
    ref = open(os.path.join(REF, "80kScene_sceneObjectDatas.json")).read().splitlines()
    assert len(mine) == len(ref)
    strip = lambda l: l.split(":")[0]   # noqa: E731  (keys, nesting and indentation; digits may differ in the last place)
    assert [strip(l) for l in mine] == [strip(l) for l in ref]


def test_errors_name_the_file(scene_small, tmp_path):
    scene_small.save_json(str(tmp_path))
    os.remove(tmp_path / "loadedVertices.json")
    with pytest.raises(rt.NativeError, match="loadedVertices.json: file not found"):
        _empty_like(scene_small).load_json(str(tmp_path))
    open(tmp_path / "loadedVertices.json", "w").write('{ "array": [ { "pos": { "x": 1.0, "y": 2.0 } } ] }')
    with pytest.raises(rt.NativeError, match=r"loadedVertices.json: missing field 'pos.z'"):
        _empty_like(scene_small).load_json(str(tmp_path))
    open(tmp_path / "loadedVertices.json", "w").write('{ "array": [ 1, 2')
    with pytest.raises(rt.NativeError, match="loadedVertices.json"):
        _empty_like(scene_small).load_json(str(tmp_path))
    open(tmp_path / "integers.json", "w").write('{ "array": [ { "Key": "maxBVHDepth", "Value": 3 } ] }')
    scene_small.save_json  # (files other than integers stay valid)
    open(tmp_path / "loadedVertices.json", "w").write('{ "array": [] }')
    with pytest.raises(rt.NativeError, match="Key 'emittingObjectsNum' not found"):
        _empty_like(scene_small).load_json(str(tmp_path))


@pytest.mark.gpu
def test_loaded_scene_renders_like_the_built_one(gpu_ctx, scene_small, tmp_path):
    scene_small.save_json(str(tmp_path))
    loaded = _empty_like(scene_small).load_json(str(tmp_path))
    cfg = h.small_config(64, 64)
    out = []
    for sc in (scene_small, loaded):
        drv = rt.Driver(gpu_ctx, cfg, sc)
        for _ in range(2):
            drv.render_frame()
        out.append((drv.texture("Result"), drv.buffer("Candidates")))
        drv.close()
    assert np.array_equal(h.bits(out[0][0]), h.bits(out[1][0])) and np.array_equal(h.bits(out[0][1]), h.bits(out[1][1]))
