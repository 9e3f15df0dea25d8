"""Host-side scene builder (host/scene_builder.cpp): mesh utilities and scene assembly."""
import numpy as np

import harness as h

rt = h.rt


def test_subdivide9_counts_and_area():
    m = rt.Mesh.load(rt.ASSET_DIR + "/suzanne.rtmesh")
    v0, t0 = m.arrays()
    s = m.subdivide9()
    v1, t1 = s.arrays()
    assert len(t1) == 9 * len(t0)
    area = lambda v, t: 0.5 * np.linalg.norm(np.cross(v["pos"][t[:, 1]] - v["pos"][t[:, 0]],
                                                      v["pos"][t[:, 2]] - v["pos"][t[:, 0]]), axis=1).sum()
    assert abs(area(v1, t1) - area(v0, t0)) < 1e-3 * area(v0, t0)      # flat subdivision keeps the surface
    # orientation is preserved: every sub-triangle's normal agrees with its parent's
    n0 = np.cross(v0["pos"][t0[:, 1]] - v0["pos"][t0[:, 0]], v0["pos"][t0[:, 2]] - v0["pos"][t0[:, 0]])
    n1 = np.cross(v1["pos"][t1[:, 1]] - v1["pos"][t1[:, 0]], v1["pos"][t1[:, 2]] - v1["pos"][t1[:, 0]])
    assert ((n1 * np.repeat(n0, 9, axis=0)).sum(axis=1) > 0).all()
    # shared edges share vertices: about 4 new vertices per triangle, not 7
    assert len(v1) < len(v0) + 4.5 * len(t0)


def test_80k_stand_in_scene(scene_80k):
    a = scene_80k.arrays()
    assert len(a["Triangles"]) == 8712 * 9 + 968
    assert a["maxBVHDepth"] == 24
    tri = a["Triangles"]
    assert (tri["area"] > 0).all() and (tri["localNormal"] == 0).all()   # normals are filled by CalcTriNormals
    assert tri["v"].min() >= 0 and tri["v"].max() < len(a["Vertices"])
    light = a["LightObjects"][0]
    assert light["brightness"] == 40000.0 and light["totTris"] == 968


def test_jitter_makes_distinct_geometry():
    m = rt.Mesh.load(rt.ASSET_DIR + "/suzanne.rtmesh")
    a, _ = m.jitter(1, 1e-4).arrays()
    b, _ = m.jitter(2, 1e-4).arrays()
    o, _ = m.arrays()
    assert np.abs(a["pos"] - o["pos"]).max() <= 1.01e-4  # amplitude, plus fp32 rounding of the sum
    assert not np.array_equal(a["pos"], b["pos"])


def test_instanced_scene_shares_bvh():
    s = rt.build_scene_instanced(grid=(3, 2), emissive_objects=2, emissive_tris=16, unique=False, subdivide=False)
    a = s.arrays()
    starts = a["SceneObjects"]["bvStartIndex"]
    assert len(a["SceneObjects"]) == 8 and len(np.unique(starts)) == 2      # one dragon BVH, one disc BVH
    assert a["emittingObjectsNum"] == 2 and len(a["LightObjects"]) == 2
    assert len(a["Triangles"]) == 8712 + 16


def test_view_space_dims_integer_aspect():
    import ctypes as C
    out = (C.c_float * 2)()
    rt.host_lib().rth_view_space_dims(70.0, 1920, 1080, out)
    assert abs(out[1] - np.tan(np.radians(35.0))) < 1e-6 and out[0] == out[1]   # 1920/1080 == 1 in integer division
    rt.host_lib().rth_view_space_dims(70.0, 3840, 1080, out)
    assert abs(out[0] - 3 * out[1]) < 1e-6


def _grid_scene():
    return rt.build_scene_instanced(grid=(3, 2), emissive_objects=2, emissive_tris=16, unique=False, subdivide=False)


def test_rebuild_after_a_move_keeps_meshes_and_appends_the_scene_bvh():
    """ConstructScene called again (ObjectHelper.cs:666-738): mesh BVHs, vertices and triangles are kept, the per-object
    data, scene BVH and light list equal those of a scene built with the object at its new place, and - the ReSTIR
    project never drops the old scene BVH (:38,613-614) - the new scene nodes sit behind the old ones."""
    pos, rot, scale = (5.0, 1.0, 20.0), (10.0, 40.0, 0.0), (8.0, 8.0, 8.0)
    moved = _grid_scene()
    before = moved.arrays()
    moved.set_object_transform(4, pos, rot, scale)
    moved.rebuild()
    after = moved.arrays()

    ref = _grid_scene()
    ref.set_object_transform(4, pos, rot, scale)
    ref.rebuild(keep_old_scene_bvh=False)      # the RIS project's builder replaces the scene BVH (MeshHelper.cs:371-372)
    want = ref.arrays()

    n_mesh_nodes = before["sceneBVHStartIndex"]
    n_scene_old = len(before["BVs"]) - n_mesh_nodes
    assert after["sceneBVHStartIndex"] == len(before["BVs"])
    assert want["sceneBVHStartIndex"] == n_mesh_nodes
    for k in ("Vertices", "Triangles"):
        assert np.array_equal(h.bits(after[k]), h.bits(before[k])), k
    assert np.array_equal(h.bits(after["BVs"][:len(before["BVs"])]), h.bits(before["BVs"]))   # mesh BVHs + the stale scene BVH
    assert np.array_equal(h.bits(after["SceneObjects"]), h.bits(want["SceneObjects"]))
    assert np.array_equal(h.bits(after["LightObjects"]), h.bits(want["LightObjects"]))
    assert after["totArea"] == want["totArea"] and after["emittingObjectsNum"] == want["emittingObjectsNum"]
    # the new scene BVH is the replaced one shifted behind the stale nodes
    new, rep = after["BVs"][after["sceneBVHStartIndex"]:], want["BVs"][n_mesh_nodes:]
    assert len(new) == len(rep)
    assert np.array_equal(new["min"], rep["min"]) and np.array_equal(new["max"], rep["max"])
    internal = rep["indexA"] < 0
    assert np.array_equal(new["indexA"][internal], rep["indexA"][internal] - n_scene_old)
    assert np.array_equal(new["indexB"][internal], rep["indexB"][internal] - n_scene_old)
    assert np.array_equal(new["indexA"][~internal], rep["indexA"][~internal])
    assert np.array_equal(new["indexB"][~internal], rep["indexB"][~internal])
    # the moved object's matrices are CreateWorldToLocalMatrix of the new transform
    w2l, l2w = rt.world_to_local(pos, rot, scale)
    assert any(np.array_equal(o["worldToLocal"], w2l) and np.array_equal(o["localToWorld"], l2w) for o in after["SceneObjects"])
    assert not np.array_equal(h.bits(after["SceneObjects"]), h.bits(before["SceneObjects"]))
