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


def test_every_file_the_reference_loader_wraps_is_written(scene_small, tmp_path):
    """LoadMultiArrayContainerFromJsonFiles (FileLoader.cs:73-87) reads twelve arrays and wraps loadedMeshes in a List:
    a missing file is a null array and the List constructor throws.  The three arrays only the reference's own scene
    construction uses are checked field by field against Triangle.cs:7-13 and MeshData.cs:6-11."""
    scene_small.save_json(str(tmp_path))
    keys = ("loadedTriangles", "sceneObjectDatas", "lightObjects", "loadedMeshesLookup", "loadedComponentDatas", "loadedMeshes",
            "loadedBVs", "loadedVertices", "integers", "floats", "renderBVs", "renderTriangles")
    assert sorted(os.listdir(tmp_path)) == sorted(f"{k}.json" for k in keys)
    data = {k: json.load(open(tmp_path / f"{k}.json"))["array"] for k in keys}
    a = scene_small.arrays()
    tris, verts = data["loadedTriangles"], a["Vertices"]["pos"]
    assert len(tris) == len(a["Triangles"]) == len(data["renderTriangles"])
    for i in (0, 1, len(tris) // 2, len(tris) - 1):
        t = tris[i]
        assert list(t) == ["vertex0Index", "vertex1Index", "vertex2Index", "parentIndex", "min", "max", "area"]
        idx = [t["vertex0Index"], t["vertex1Index"], t["vertex2Index"]]
        assert idx == [int(v) for v in a["Triangles"][i]["v"]] and t["parentIndex"] == 0
        assert np.array_equal(np.array([t["min"][c] for c in "xyz"], np.float32), verts[idx].min(axis=0))
        assert np.array_equal(np.array([t["max"][c] for c in "xyz"], np.float32), verts[idx].max(axis=0))
        assert np.float32(t["area"]) == a["Triangles"][i]["area"]
    # one mesh per distinct mesh BVH, in order of first use; every scene object points at its own
    lookup, meshes = data["loadedMeshesLookup"], data["loadedMeshes"]
    assert len(lookup) == len(a["SceneObjects"]) and sorted(set(lookup)) == list(range(len(meshes)))
    total = 0
    for o, m in zip(a["SceneObjects"], lookup):
        mesh = meshes[m]
        assert list(mesh) == ["triangles", "localMin", "localMax", "componentStartIndex", "bvStartIndex", "meshKey"]
        root = int(o["bvStartIndex"])
        assert mesh["bvStartIndex"] == root
        start, count = (int(v) for v in a["ComponentDatas"][root])
        assert mesh["componentStartIndex"] == start and len(mesh["triangles"]) == count
        assert mesh["triangles"][0] == tris[start] and mesh["triangles"][-1] == tris[start + count - 1]
        assert np.array_equal(np.array([mesh["localMin"][c] for c in "xyz"], np.float32), a["BVs"][root]["min"])
        assert np.array_equal(np.array([mesh["localMax"][c] for c in "xyz"], np.float32), a["BVs"][root]["max"])
        assert isinstance(mesh["meshKey"], str) and mesh["meshKey"]
    for mesh in meshes:
        total += len(mesh["triangles"])
    assert total == len(tris)
    # the layout of the text is JsonUtility's pretty print: the committed lookup file of the reference, byte for byte
    assert open(tmp_path / "loadedMeshesLookup.json").read() == '{\n    "array": [\n        0,\n        1\n    ]\n}'


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
