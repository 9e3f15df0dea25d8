#!/usr/bin/env python3
"""Converts a Wavefront OBJ mesh into the small binary .rtmesh format the host-side scene builder loads.

Run at development time only (the GPU box has no reference checkout); the outputs are committed under assets/.

    python tools/import_obj.py /root/reference/BVH_ReSTIR_DL/Assets/OBJs/Dragon8kUV.obj assets/dragon8k.rtmesh

Import convention (Unity's own OBJ importer cannot be reproduced: it re-indexes and welds vertices in an
undocumented order): one vertex per distinct (position index, uv index) pair in order of first appearance,
faces fan-triangulated; the mesh is converted from OBJ's right-handed frame to Unity's left-handed one by
negating x and reversing the winding, so cross(v1-v0, v2-v0) is the outward normal the kernels' back-face
test expects (BVH_ReSTIR_DL/Assets/Scripts/ComputeShaders/ReSTIR_BVH_RayTracer.compute:182).

.rtmesh layout (little endian): 8-byte magic "RTMESH1\0", int32 vertex count, int32 triangle count,
vertices as 5 float32 (pos.xyz, uv.xy), triangles as 3 int32.
"""
import struct
import sys

import numpy as np


def load_obj(path):
    pos, uvs = [], []
    verts, index, tris = [], {}, []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                pos.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "vt":
                uvs.append((float(t[1]), float(t[2])))
            elif t[0] == "f":
                corner = []
                for c in t[1:]:
                    parts = c.split("/")
                    vi = int(parts[0]) - 1
                    ti = int(parts[1]) - 1 if len(parts) > 1 and parts[1] else -1
                    key = (vi, ti)
                    if key not in index:
                        index[key] = len(verts)
                        p = pos[vi]
                        uv = uvs[ti] if ti >= 0 else (-1.0, -1.0)
                        verts.append((-p[0], p[1], p[2], uv[0], uv[1]))
                    corner.append(index[key])
                for k in range(1, len(corner) - 1):
                    tris.append((corner[0], corner[k + 1], corner[k]))  # reversed winding
    return np.asarray(verts, dtype=np.float32), np.asarray(tris, dtype=np.int32)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    v, t = load_obj(src)
    with open(dst, "wb") as f:
        f.write(b"RTMESH1\0")
        f.write(struct.pack("<ii", len(v), len(t)))
        f.write(v.tobytes())
        f.write(t.tobytes())
    print(f"{dst}: {len(v)} vertices, {len(t)} triangles, bbox {v[:, :3].min(0)} .. {v[:, :3].max(0)}")


if __name__ == "__main__":
    main()
