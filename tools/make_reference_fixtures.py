#!/usr/bin/env python3
"""Extracts the numbers of the reference's committed scene dump (BVH_ReSTIR_DL/80kScene/*.json) into one small
fixture, tests/golden/ref_80kscene.json.gz, so tests can pin the host-side scene builder against the reference's
own data without reading the reference checkout at test time.

    python tools/make_reference_fixtures.py /root/reference

Only data is copied (matrices, AABBs, integer tables); no source text.
"""
import gzip
import json
import os
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
d = os.path.join(ref, "BVH_ReSTIR_DL", "80kScene")


def load(name):
    with open(os.path.join(d, f"80kScene_{name}.json")) as f:
        return json.load(f)["array"]


def mat(m):
    return [m[c][k] for c in ("c0", "c1", "c2", "c3") for k in ("x", "y", "z", "w")]  # column-major, 16 floats


objs = []
for o in load("sceneObjectDatas"):
    objs.append({"worldToLocal": mat(o["worldToLocalMatrix"]), "localToWorld": mat(o["localToWorldMatrix"]),
                 "materialIndex": o["materialIndex"], "bvStartIndex": o["bvStartIndex"], "maxDepthBVH": o["maxDepthBVH"],
                 "areaApprox": o["areaApprox"], "min": [o["min"][k] for k in "xyz"], "max": [o["max"][k] for k in "xyz"]})
lights = [{"localToWorld": mat(l["localToWorldMatrix"]), "areaApprox": l["areaApprox"], "brightness": l["brightness"],
           "triStart": l["triStart"], "totTris": l["totTris"]} for l in load("lightObjects")]
comp = [[c["x"], c["y"]] for c in load("loadedComponentDatas")]
out = {"source": "BVH_ReSTIR_DL/80kScene/80kScene_*.json of Lundalex/Ray-Tracer @ 2024-10-08",
       "sceneObjectDatas": objs, "lightObjects": lights,
       "integers": {e["Key"]: e["Value"] for e in load("integers")},
       "floats": {e["Key"]: e["Value"] for e in load("floats")},
       "loadedMeshesLookup": load("loadedMeshesLookup"),
       "loadedComponentDatas": comp,
       "renderTrianglesCount": len(load("renderTriangles"))}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_80kscene.json.gz")
with gzip.open(dst, "wt") as f:
    json.dump(out, f)
print(dst, os.path.getsize(dst), "bytes;", len(comp), "component pairs")
