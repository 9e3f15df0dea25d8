"""Known-answer tests that pin the oracle and the host-side scene builder.

What the reference offers to pin against: no test vectors at all for the arithmetic (it has no tests), but a
committed dump of one built scene (BVH_ReSTIR_DL/80kScene/*.json, extracted into tests/golden/ref_80kscene.json.gz
by tools/make_reference_fixtures.py).  That dump pins the transform code and the BVH topology of the builder; the
RNG is pinned against an independent big-integer restatement written here; intersection routines against
hand-derived cases.
"""
import ctypes as C
import gzip
import json
import math
import os

import numpy as np
import pytest

import harness as h

rt = h.rt
lib = h.oracle_lib()
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref_dump():
    with gzip.open(os.path.join(GOLDEN, "ref_80kscene.json.gz"), "rt") as f:
        return json.load(f)


# ---------------------------------------------------------------- RNG (RendererResources.hlsl:63-91)
def pcg_reference(state):
    """Independent restatement with Python integers."""
    state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
    result = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    result = ((result >> 22) ^ result) & 0xFFFFFFFF
    return state, result


@pytest.mark.parametrize("seed", [0, 1, 12345, 2073599 + 999998, 0xFFFFFFFF])
def test_next_random_sequence(seed):
    s_ref = seed
    s = C.c_uint32(seed)
    for _ in range(8):
        s_ref, want = pcg_reference(s_ref)
        got = lib.orc_next_random(C.byref(s))
        assert got == want and s.value == s_ref


def test_rand_normalized_can_reach_one():
    # u >= 0xFFFFFF80 rounds to 2^32 as a float, so u / 2^32 is exactly 1.0 (and randIntSpan can return b)
    for u in (0xFFFFFF80, 0xFFFFFFFF):
        assert np.float32(u) / np.float32(4294967296.0) == np.float32(1.0)
    assert np.float32(0xFFFFFF7F) / np.float32(4294967296.0) < np.float32(1.0)
    s = C.c_uint32(99)
    for _ in range(2000):
        st = s.value
        r = lib.orc_rand_normalized(C.byref(s))
        _, want_u = pcg_reference(st)
        assert r == float(np.float32(want_u) / np.float32(4294967296.0))
        assert 0.0 <= r <= 1.0


def test_rand_int_span_and_weighted_rand():
    s = C.c_uint32(4242)
    for _ in range(2000):
        st = s.value
        v = lib.orc_rand_int_span(-27, 27, C.byref(s))
        _, u = pcg_reference(st)
        r = np.float32(u) / np.float32(4294967296.0)
        assert v == -27 + int(np.float32(54) * r)
        assert -27 <= v <= 27
    # weightedRand(a, b): r*b < a, with b the total BEFORE adding the candidate (ReSTIR variant)
    s = C.c_uint32(1)
    assert lib.orc_weighted_rand(0.0, 0.0, C.byref(s)) == 0       # 0 < 0 is false
    assert lib.orc_weighted_rand(1.0, 0.0, C.byref(s)) == 1       # first candidate always wins
    st = s.value
    _, u = pcg_reference(st)
    r = np.float32(u) / np.float32(4294967296.0)
    assert lib.orc_weighted_rand(3.0, 10.0, C.byref(s)) == int(np.float32(r * np.float32(10.0)) < np.float32(3.0))


# ---------------------------------------------------------------- intersection routines
def f3(*v):
    return (C.c_float * 3)(*v)


def bv(mn, mx):
    a = np.zeros(1, rt.DTYPES["BV"])
    a["min"], a["max"] = mn, mx
    return a


def test_ray_box():
    box = bv((1, -1, -1), (3, 1, 1))
    hit = lambda p, d: lib.orc_ray_bv_intersect(f3(*p), f3(*d), box.ctypes.data)
    assert hit((0, 0, 0), (1, 0, 0)) == 1.0                       # entry distance
    assert hit((2, 0, 0), (1, 0, 0)) == 0.0                       # origin inside: 0
    assert hit((5, 0, 0), (1, 0, 0)) == math.inf                  # box behind the ray
    assert hit((0, 5, 0), (1, 0, 0)) == math.inf                  # parallel miss; 0*inf = NaN is ignored by min/max
    assert hit((0, 0.5, 0), (1, 0, 0)) == 1.0                     # zero direction components inside the slab
    assert hit((0, 0, 0), (0.6, 0.8, 0)) == math.inf              # leaves through the top before x = 1
    d = np.float32(1) / np.float32(0.6)
    assert abs(hit((0, -2, 0), (0.6, 0.8, 0)) - float(np.float32(1.0) * d)) < 1e-6


def vert(p, uv=(0, 0)):
    a = np.zeros(1, rt.DTYPES["Vertex"])
    a["pos"], a["uv"] = p, uv
    return a


def test_ray_triangle():
    # cross(v1-v0, v2-v0) = +z, so rays travelling towards -z see the front face (determinant > 0)
    v0, v1, v2 = vert((0, 0, 0), (0, 0)), vert((1, 0, 0), (1, 0)), vert((0, 1, 0), (0, 1))
    dst, uv = C.c_float(), (C.c_float * 2)()
    call = lambda p, d: lib.orc_ray_triangle_intersect(f3(*p), f3(*d), v0.ctypes.data, v1.ctypes.data, v2.ctypes.data,
                                                       C.byref(dst), uv)
    assert call((0.25, 0.25, 1), (0, 0, -1)) == 1 and dst.value == 1.0
    assert (uv[0], uv[1]) == (0.25, 0.25)                          # uv = v0.uv*w + v1.uv*u + v2.uv*v
    assert call((0.25, 0.25, -1), (0, 0, 1)) == 0                  # back face: determinant < 1e-8
    assert call((0.75, 0.75, 1), (0, 0, -1)) == 0                  # outside (w < 0)
    assert call((0.25, 0.25, -1), (0, 0, -1)) == 0                 # behind the origin (dst < 0)
    assert call((0.25, 0.25, 1), (1, 0, 0)) == 0                   # edge-on: determinant 0 -> inf/NaN compare false
    assert call((0, 0, 1), (0, 0, -1)) == 1                        # exactly on a vertex: u = v = 0, w = 1


def test_brdf_x_channel():
    # GetBRDF returns the x channel of a float3 expression: with metallicity forced to 1 and albedo.x = 1 the
    # diffuse term vanishes and F0.x = 1, so F.x = 1 and the result is D*G/(4*NdotL*NdotV+0.001)*NdotL*10.
    n = (0.0, 0.0, 1.0)
    incoming = (0.0, 0.0, -1.0)      # arriving straight down; the routine negates it
    outgoing = (0.0, 0.0, 1.0)
    r = 0.8
    got = lib.orc_get_brdf(f3(*incoming), f3(*outgoing), f3(*n), 0.3, r)
    a2 = (r * r) ** 2
    D = a2 / (3.14 * (1 * (a2 - 1) + 1) ** 2)
    k = (r + 1) ** 2 / 8
    G = (1 / (1 * (1 - k) + k)) ** 2
    want = D * G * 1.0 / (4 * 1 * 1 + 0.001) * 1 * 10
    assert abs(got - want) < 1e-5 * want


# ---------------------------------------------------------------- pinned by the reference's scene dump
def test_world_to_local_matches_dump(ref_dump):
    """Utils.CreateWorldToLocalMatrix (Resources.cs:67-87) reproduces the matrices Unity wrote to the dump.
    Transforms from Ray Tracer.unity:468-506,785-822 and Suzanne1k.prefab:16-25."""
    dragon, suzanne = ref_dump["sceneObjectDatas"]
    w2l, l2w = rt.world_to_local((-4, -2, 13), (0, 0, 0), (12, 12, 12))
    np.testing.assert_allclose(w2l, dragon["worldToLocal"], rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(l2w, dragon["localToWorld"], rtol=2e-7, atol=1e-6)
    w2l, l2w = rt.world_to_local((0.5, 4.4, 14.6), (0, 190, 0), (2, 2, 2))
    np.testing.assert_allclose(w2l, suzanne["worldToLocal"], rtol=3e-6, atol=2e-6)
    np.testing.assert_allclose(l2w, suzanne["localToWorld"], rtol=3e-6, atol=2e-6)


def test_builder_reproduces_dump_topology(ref_dump, scene_small):
    """The C++ builder, run on Suzanne + the 8.7k dragon with the dump's parameters (SplitResolution 5, depths 14
    and 18), reproduces the dump's BVH: same node count, same (componentStart, totComponents) for every one of the
    18,525 nodes, same scene-object order, same light object.  The split search depends only on the triangles'
    bounding boxes, so Unity's different vertex indexing does not matter."""
    a = scene_small.arrays()
    assert a["sceneBVHStartIndex"] == ref_dump["integers"]["sceneBVHStartIndex"] == 18522
    assert a["maxBVHDepth"] == ref_dump["integers"]["maxBVHDepth"]
    assert a["emittingObjectsNum"] == ref_dump["integers"]["emittingObjectsNum"]
    assert len(a["Triangles"]) == ref_dump["renderTrianglesCount"] == 9680
    comp = np.stack([a["ComponentDatas"]["x"], a["ComponentDatas"]["y"]], axis=1)
    want = np.asarray(ref_dump["loadedComponentDatas"], dtype=np.int32)
    assert comp.shape == want.shape
    same = (comp == want).all(axis=1)
    assert same.all(), f"{(~same).sum()} of {len(same)} BVH nodes differ, first at {int(np.argmin(same))}"
    for got, ref in zip(a["SceneObjects"], ref_dump["sceneObjectDatas"]):
        assert got["bvStartIndex"] == ref["bvStartIndex"] and got["materialIndex"] == ref["materialIndex"]
        assert got["maxDepthBVH"] == ref["maxDepthBVH"]
        np.testing.assert_allclose(got["min"], ref["min"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(got["max"], ref["max"], rtol=1e-5, atol=1e-5)
        assert abs(got["areaApprox"] - ref["areaApprox"]) < 1e-3 * ref["areaApprox"]
    assert abs(a["totArea"] - ref_dump["floats"]["totArea"]) < 1e-3
    light, ref_light = a["LightObjects"][0], ref_dump["lightObjects"][0]
    assert (light["triStart"], light["totTris"]) == (ref_light["triStart"], ref_light["totTris"]) == (0, 968)
    np.testing.assert_allclose(light["localToWorld"], ref_light["localToWorld"], rtol=3e-6, atol=2e-6)


def test_bvh_invariants(scene_small):
    """DFS-preorder layout (childA = parent + 1), leaves partition the triangle range, boxes contain children."""
    a = scene_small.arrays()
    bvs, n = a["BVs"], len(a["BVs"])
    for start in {int(o["bvStartIndex"]) for o in a["SceneObjects"]} | {a["sceneBVHStartIndex"]}:
        covered, stack = [], [start]
        while stack:
            i = stack.pop()
            b = bvs[i]
            if b["indexA"] >= 0:
                covered.append((int(b["indexA"]), int(b["indexB"])))
            else:
                ca, cb = -int(b["indexA"]), -int(b["indexB"])
                assert ca == i + 1 and i < cb < n
                for c in (ca, cb):
                    assert (bvs[c]["min"] >= b["min"] - 1e-6).all() and (bvs[c]["max"] <= b["max"] + 1e-6).all()
                stack += [cb, ca]
        covered.sort()
        pos = covered[0][0]
        for s, c in covered:
            assert s == pos and c >= 1
            pos += c
