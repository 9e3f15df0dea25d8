// scene_builder.cpp — host-side scene and BVH construction (C++ restatement of the reference's C# builder).
//
// Produces the flat structured buffers the kernels consume, in the reference's layouts and orders:
//   ObjectHelper.ConstructScene       BVH_ReSTIR_DL/Assets/Scripts/ObjectHelper.cs:666-738
//   ObjectHelper.RecursivelySplitBV   :219-290   (binned split search, DFS-preorder node array)
//   ObjectHelper.ConstructBVH         :292-352
//   ObjectHelper.ConstructSceneBVH    :609-664
//   Utils.CreateWorldToLocalMatrix    BVH_ReSTIR_DL/Assets/Scripts/Resources.cs:67-87
//   Triangle.Initialise               BVH_ReSTIR_DL/Assets/Scripts/Data types/Other/Triangle.cs:17-46
// CPU only; no GPU work happens here.
#include "rth_api.h"
#include "scene_internal.h"
#include "unity_math.h"

#include <omp.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <array>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace {

struct V3 { float x, y, z; };
inline V3 vmin(V3 a, V3 b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
inline V3 vmax(V3 a, V3 b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }
inline V3 toV3(rt_float3 v) { return {v.x, v.y, v.z}; }
inline rt_float3 toR3(V3 v) { return {v.x, v.y, v.z}; }

}  // namespace

struct rth_mesh_s {
    std::vector<rt_vertex> verts;
    std::vector<int32_t> tris;  // 3 per triangle
};

// ----------------------------------------------------------------------------------------------- meshes
extern "C" {

rth_mesh rth_mesh_create(const rt_vertex* verts, int nverts, const int32_t* tris, int ntris) {
    if (!verts || !tris || nverts <= 0 || ntris <= 0) return nullptr;
    for (int i = 0; i < 3 * ntris; i++)
        if (tris[i] < 0 || tris[i] >= nverts) return nullptr;
    rth_mesh m = new rth_mesh_s;
    m->verts.assign(verts, verts + nverts);
    m->tris.assign(tris, tris + 3 * (size_t)ntris);
    return m;
}

rth_mesh rth_mesh_load(const char* path) {
    FILE* f = path ? fopen(path, "rb") : nullptr;
    if (!f) return nullptr;
    char magic[8];
    int32_t nv = 0, nt = 0;
    rth_mesh m = nullptr;
    if (fread(magic, 1, 8, f) == 8 && memcmp(magic, "RTMESH1\0", 8) == 0 && fread(&nv, 4, 1, f) == 1 &&
        fread(&nt, 4, 1, f) == 1 && nv > 0 && nt > 0 && nv < (1 << 28) && nt < (1 << 28)) {
        std::vector<rt_vertex> v((size_t)nv);
        std::vector<int32_t> t((size_t)nt * 3);
        if (fread(v.data(), sizeof(rt_vertex), (size_t)nv, f) == (size_t)nv &&
            fread(t.data(), 4, (size_t)nt * 3, f) == (size_t)nt * 3)
            m = rth_mesh_create(v.data(), nv, t.data(), nt);
    }
    fclose(f);
    return m;
}

rth_mesh rth_mesh_subdivide9(rth_mesh src) {
    if (!src) return nullptr;
    rth_mesh m = new rth_mesh_s;
    m->verts = src->verts;
    // Two new vertices per undirected edge, at 1/3 and 2/3 from the lower-indexed end, so both triangles that
    // share the edge get the very same vertices.
    std::map<std::pair<int, int>, int> edgeFirst;
    auto lerpV = [](const rt_vertex& a, const rt_vertex& b, float t) {
        rt_vertex r;
        r.pos = {a.pos.x + (b.pos.x - a.pos.x) * t, a.pos.y + (b.pos.y - a.pos.y) * t, a.pos.z + (b.pos.z - a.pos.z) * t};
        r.uv = {a.uv.x + (b.uv.x - a.uv.x) * t, a.uv.y + (b.uv.y - a.uv.y) * t};
        return r;
    };
    // returns the index of the point at `third`/3 of the way from a to b (third = 1 or 2)
    auto edgePoint = [&](int a, int b, int third) {
        int lo = a < b ? a : b, hi = a < b ? b : a;
        auto key = std::make_pair(lo, hi);
        auto it = edgeFirst.find(key);
        int base;
        if (it == edgeFirst.end()) {
            base = (int)m->verts.size();
            edgeFirst[key] = base;
            m->verts.push_back(lerpV(src->verts[lo], src->verts[hi], 1.0f / 3.0f));
            m->verts.push_back(lerpV(src->verts[lo], src->verts[hi], 2.0f / 3.0f));
        } else {
            base = it->second;
        }
        int fromLo = (a == lo) ? third : 3 - third;
        return base + (fromLo - 1);
    };
    size_t nt = src->tris.size() / 3;
    m->tris.reserve(nt * 27);
    for (size_t t = 0; t < nt; t++) {
        int a = src->tris[3 * t], b = src->tris[3 * t + 1], c = src->tris[3 * t + 2];
        int ab1 = edgePoint(a, b, 1), ab2 = edgePoint(a, b, 2);
        int bc1 = edgePoint(b, c, 1), bc2 = edgePoint(b, c, 2);
        int ca1 = edgePoint(c, a, 1), ca2 = edgePoint(c, a, 2);
        int ctr = (int)m->verts.size();
        {
            const rt_vertex &A = src->verts[a], &B = src->verts[b], &C = src->verts[c];
            rt_vertex g;
            g.pos = {(A.pos.x + B.pos.x + C.pos.x) / 3.0f, (A.pos.y + B.pos.y + C.pos.y) / 3.0f,
                     (A.pos.z + B.pos.z + C.pos.z) / 3.0f};
            g.uv = {(A.uv.x + B.uv.x + C.uv.x) / 3.0f, (A.uv.y + B.uv.y + C.uv.y) / 3.0f};
            m->verts.push_back(g);
        }
        const int out[9][3] = {{a, ab1, ca2},   {ab1, ctr, ca2}, {ab1, ab2, ctr}, {ab2, bc1, ctr}, {ab2, b, bc1},
                               {ca2, ctr, ca1}, {ctr, bc2, ca1}, {ctr, bc1, bc2}, {ca1, bc2, c}};
        for (auto& o : out) { m->tris.push_back(o[0]); m->tris.push_back(o[1]); m->tris.push_back(o[2]); }
    }
    return m;
}

rth_mesh rth_mesh_jitter(rth_mesh src, uint32_t seed, float amplitude) {
    if (!src) return nullptr;
    rth_mesh m = new rth_mesh_s(*src);
    uint32_t s = seed * 2654435761u + 12345u;
    auto next = [&]() {
        s = s * 747796405u + 2891336453u;
        uint32_t r = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u;
        r = (r >> 22) ^ r;
        return ((float)r / 4294967296.0f) * 2.0f - 1.0f;
    };
    for (auto& v : m->verts) {
        v.pos.x += next() * amplitude;
        v.pos.y += next() * amplitude;
        v.pos.z += next() * amplitude;
    }
    return m;
}

rth_mesh rth_mesh_disc(int ntris) {
    if (ntris < 1) return nullptr;
    rth_mesh m = new rth_mesh_s;
    m->verts.push_back({{0, 0, 0}, {0.5f, 0.5f}});
    int rim = ntris < 3 ? 3 : ntris;
    for (int i = 0; i < rim; i++) {
        double a = 6.283185307179586 * i / rim;
        float cx = (float)cos(a), cy = (float)sin(a);
        m->verts.push_back({{cx, cy, 0}, {0.5f + 0.5f * cx, 0.5f + 0.5f * cy}});
    }
    for (int i = 0; i < ntris; i++) {
        // cross(v1-v0, v2-v0) = +z for this order (left-handed convention: see tools/import_obj.py)
        m->tris.push_back(0);
        m->tris.push_back(1 + i % rim);
        m->tris.push_back(1 + (i + 1) % rim);
    }
    return m;
}

// The same disc tessellated as `sectors` wedges x rings: a fan of `sectors` triangles in the middle, two triangles per
// sector in every further ring, i.e. sectors * (2 * rings - 1) triangles of similar size (a fan of many triangles is
// all slivers through one vertex, whose boxes overlap: every ray near the disc then tests dozens of them).  `ntris`
// is split as sectors * (2 * rings - 1) with the largest odd factor not above sqrt(2 * ntris); a count without such
// a factor above 1 gives the plain fan.
rth_mesh rth_mesh_ring_disc(int ntris) {
    if (ntris < 1) return nullptr;
    int odd = 1;
    for (int d = 3; (long long)d * d <= 2LL * ntris; d += 2) if (ntris % d == 0) odd = d;
    const int rings = (odd + 1) / 2, sectors = ntris / odd;
    if (rings < 2 || sectors < 3) return rth_mesh_disc(ntris);
    rth_mesh m = new rth_mesh_s;
    m->verts.push_back({{0, 0, 0}, {0.5f, 0.5f}});
    for (int k = 1; k <= rings; k++)
        for (int i = 0; i < sectors; i++) {
            double a = 6.283185307179586 * i / sectors, r = (double)k / rings;
            float cx = (float)(r * cos(a)), cy = (float)(r * sin(a));
            m->verts.push_back({{cx, cy, 0}, {0.5f + 0.5f * cx, 0.5f + 0.5f * cy}});
        }
    auto v = [&](int ring, int i) { return 1 + (ring - 1) * sectors + i % sectors; };
    for (int i = 0; i < sectors; i++) { m->tris.push_back(0); m->tris.push_back(v(1, i)); m->tris.push_back(v(1, i + 1)); }
    for (int k = 2; k <= rings; k++)
        for (int i = 0; i < sectors; i++) {  // both halves of the quad face +z like the fan
            m->tris.push_back(v(k - 1, i)); m->tris.push_back(v(k, i)); m->tris.push_back(v(k, i + 1));
            m->tris.push_back(v(k - 1, i)); m->tris.push_back(v(k, i + 1)); m->tris.push_back(v(k - 1, i + 1));
        }
    return m;
}

void rth_mesh_free(rth_mesh m) { delete m; }
int rth_mesh_counts(rth_mesh m, int* nverts, int* ntris) {
    if (!m) return -1;
    if (nverts) *nverts = (int)m->verts.size();
    if (ntris) *ntris = (int)(m->tris.size() / 3);
    return 0;
}
const rt_vertex* rth_mesh_vertices(rth_mesh m) { return m ? m->verts.data() : nullptr; }
const int32_t* rth_mesh_triangles(rth_mesh m) { return m ? m->tris.data() : nullptr; }

}  // extern "C"

// ----------------------------------------------------------------------------------------------- BVH builder
namespace {

// One IBVHComponent (Data types/Other/BVHComponent.cs): an AABB plus the index of the payload it stands for.
struct Comp {
    V3 mn, mx;
    int32_t payload;
};
struct BVNode {  // Data types/Other/BV.cs
    V3 mn, mx;
    int componentStart, totComponents, childIndexA, childIndexB;
};

// Func.Avg(min, max) (Resources.cs:335-345): tot = 0; tot += min; tot += max; tot / 2.
inline V3 getMid(const Comp& c) {
    V3 tot = {0.0f, 0.0f, 0.0f};
    tot = {tot.x + c.mn.x, tot.y + c.mn.y, tot.z + c.mn.z};
    tot = {tot.x + c.mx.x, tot.y + c.mx.y, tot.z + c.mx.z};
    return {tot.x / 2.0f, tot.y / 2.0f, tot.z / 2.0f};
}
inline float axisOf(V3 v, int axis) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; }
// ObjectHelper.GetBoxArea (:149-158)
inline float boxArea(V3 a, V3 b) {
    float length = fabsf(a.x - b.x), width = fabsf(a.y - b.y), height = fabsf(a.z - b.z);
    return 2 * (length * width + width * height + height * length);
}

// The same expression as the reference's runtime evaluates it for the value it STORES per scene object: the committed
// 80kScene dump holds 422.67606 for the dragon where single-precision evaluation gives 422.6761, i.e. the products
// and sums were carried in double precision and rounded once (Mono keeps float expressions in double registers).
inline float boxAreaStored(V3 a, V3 b) {
    double length = fabsf(a.x - b.x), width = fabsf(a.y - b.y), height = fabsf(a.z - b.z);
    return (float)(2 * (length * width + width * height + height * length));
}

struct Builder {
    int splitResolution;
    std::vector<BVNode> nodes;
    std::vector<Comp>* comps;

    // RecursivelySplitBV (:219-290).  `depth` counts from 0 at the call for the root.
    int split(int parentIndex, int maxDepth, int depth) {
        depth += 1;
        if (depth >= maxDepth) { setLeaf(parentIndex); return parentIndex; }
        const BVNode parent = nodes[parentIndex];
        int bestAxis = -1;
        V3 bestCoord = {0, 0, 0};
        float bestCost = FLT_MAX;
        V3 diff = {parent.mx.x - parent.mn.x, parent.mx.y - parent.mn.y, parent.mx.z - parent.mn.z};
        const Comp* c = comps->data() + parent.componentStart;
        const int n = parent.totComponents;
        for (int s = 0; s < splitResolution; s++) {
            float k = (float)s + 0.5f;
            V3 splitCoord = {parent.mn.x + diff.x * k / (float)splitResolution,
                             parent.mn.y + diff.y * k / (float)splitResolution,
                             parent.mn.z + diff.z * k / (float)splitResolution};
            for (int axis = 0; axis < 3; axis++) {
                // DivideIntoSubGroupsCopy (:197-217) + GetCost (:160-167), fused into one pass
                V3 mnA = {FLT_MAX, FLT_MAX, FLT_MAX}, mxA = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, mnB = mnA, mxB = mxA;
                int countA = 0;
                float sc = axisOf(splitCoord, axis);
                for (int i = 0; i < n; i++) {
                    if (axisOf(getMid(c[i]), axis) < sc) { countA++; mnA = vmin(mnA, c[i].mn); mxA = vmax(mxA, c[i].mx); }
                    else { mnB = vmin(mnB, c[i].mn); mxB = vmax(mxB, c[i].mx); }
                }
                // An empty side has an infinite box area times a zero count = NaN, which never compares below
                // the best cost, so such a split is never taken.
                if (countA == 0 || countA == n) continue;
                float cost = boxArea(mnA, mxA) * (float)countA + boxArea(mnB, mxB) * (float)(n - countA);
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestCoord = splitCoord; }
            }
        }
        if (bestAxis == -1) { setLeaf(parentIndex); return parentIndex; }

        // DivideIntoSubGroupsRef (:171-195): group A is moved to the front by forward swaps.
        Comp* cw = comps->data();
        int highestIndexA = parent.componentStart - 1, countA = 0;
        float sc = axisOf(bestCoord, bestAxis);
        for (int i = parent.componentStart; i < parent.componentStart + n; i++) {
            if (axisOf(getMid(cw[i]), bestAxis) < sc) {
                highestIndexA++;
                countA++;
                if (highestIndexA != i) std::swap(cw[highestIndexA], cw[i]);
            }
        }
        int furthest = parentIndex;
        if (countA != 0) {
            int childA = parentIndex + 1;
            nodes[parentIndex].childIndexA = childA;
            nodes.push_back(makeNode(parent.componentStart, countA));
            furthest = split(childA, maxDepth, depth);
        }
        if (n - countA != 0) {
            int childB = furthest + 1;
            nodes[parentIndex].childIndexB = childB;
            nodes.push_back(makeNode(parent.componentStart + countA, n - countA));
            furthest = split(childB, maxDepth, depth);
        }
        return furthest;
    }
    void setLeaf(int i) { nodes[i].childIndexA = -1; nodes[i].childIndexB = -1; }
    BVNode makeNode(int start, int count) const {
        V3 mn = {FLT_MAX, FLT_MAX, FLT_MAX}, mx = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = start; i < start + count; i++) { mn = vmin(mn, (*comps)[i].mn); mx = vmax(mx, (*comps)[i].mx); }
        return {mn, mx, start, count, -1, -1};
    }
    // ConstructBVH (:296-301) / ConstructSceneBVH (:639-644): the root starts with children (1, 2).
    void build(std::vector<Comp>& components, int maxDepth) {
        comps = &components;
        nodes.clear();
        BVNode root = makeNode(0, (int)components.size());
        root.childIndexA = 1;
        root.childIndexB = 2;
        nodes.push_back(root);
        split(0, maxDepth, 0);
    }
};

// BV.ClassToStruct (Data types/Other/BV.cs:20-38)
inline rt_bv toRenderBV(const BVNode& b) {
    bool isLeaf = b.childIndexA == -1 && b.childIndexB == -1;
    rt_bv r;
    r.min = toR3(b.mn);
    r.max = toR3(b.mx);
    r.indexA = isLeaf ? b.componentStart : -b.childIndexA;
    r.indexB = isLeaf ? b.totComponents : -b.childIndexB;
    return r;
}

// Func.Mul (Resources.cs:356-360): math.mul(float4x4, float4(a, 1)).xyz = c0*x + c1*y + c2*z + c3*1
inline V3 mulPoint(const M4& M, V3 a) {
    const float* m = M.m;
    return {((m[0] * a.x + m[4] * a.y) + m[8] * a.z) + m[12] * 1.0f, ((m[1] * a.x + m[5] * a.y) + m[9] * a.z) + m[13] * 1.0f,
            ((m[2] * a.x + m[6] * a.y) + m[10] * a.z) + m[14] * 1.0f};
}

struct ObjectDesc {
    int meshId;
    float pos[3], rot[3], scale[3];
    int materialIndex, maxDepthBVH;
};
struct LoadedMesh {  // Data types/Other/MeshData.cs
    V3 localMin, localMax;
    int componentStartIndex, bvStartIndex;
    bool built = false;
};

}  // namespace

struct rth_scene_s {
    int splitResolution, maxDepthSceneBVH;
    std::vector<rth_mesh_s> meshes;
    std::vector<rth_material_input> materialInputs;
    struct Tex { int w = 0, h = 0; std::vector<unsigned char> px; };
    std::vector<std::array<Tex, 5>> materialTextures;  // per material, indexed by RTH_TEX_*
    std::vector<unsigned char> atlas;
    int atlasW = 1, atlasH = 1;
    std::vector<ObjectDesc> objects;
    std::vector<LoadedMesh> loaded;  // LoadedMeshes (ObjectHelper.cs:30): the meshes whose BVH is part of BVs, by mesh id
    // outputs
    std::vector<rt_bv> BVs;
    std::vector<rt_int2> componentDatas;
    std::vector<rt_vertex> vertices;
    std::vector<rt_triangle> triangles;
    std::vector<rt_scene_object> sceneObjects;
    std::vector<rt_light_object> lightObjects;
    std::vector<rt_material> materials;
    int maxBVHDepth = 0, emittingObjectsNum = 0, sceneBVHStartIndex = 0;
    int lastSceneBVHLength = 0;  // nodes of the scene BVH at the end of BVs
    float totArea = 0;
    bool built = false;
};

namespace {

// The part of ConstructScene that runs on every call (ObjectHelper.cs:684-712): LoadSceneObjects' per-object data
// (:561-602, the mesh BVHs themselves are looked up by mesh key and kept), ConstructSceneBVH (:609-664) and the light
// list (:694-712).  The scene BVH is APPENDED to BVs / componentDatas: the reference means to drop the previous one
// first (:613-614) but never records its length (LastSceneBVHLength stays 0, :38), so after a reload the old scene
// nodes stay in the buffer, unreferenced, and SceneBVHStartIndex moves up past them.
void constructSceneLevel(rth_scene s) {
    const int nObj = (int)s->objects.size();
    const std::vector<LoadedMesh>& loaded = s->loaded;
    std::vector<rt_scene_object> objs(nObj);
    int maxBVHDepth = s->maxDepthSceneBVH;
    double totArea = 0;  // rounded once at the end, see boxAreaStored
    for (int i = 0; i < nObj; i++) {
        const ObjectDesc& o = s->objects[i];
        rt_scene_object d;
        memset(&d, 0, sizeof d);
        M4 w2l = createWorldToLocal(o.pos, o.rot, o.scale);
        M4 l2w = inverse(w2l);
        memcpy(d.worldToLocalMatrix.m, w2l.m, 64);
        memcpy(d.localToWorldMatrix.m, l2w.m, 64);
        d.materialIndex = o.materialIndex;
        d.maxDepthBVH = o.maxDepthBVH;
        d.bvStartIndex = loaded[o.meshId].bvStartIndex;
        if (o.maxDepthBVH > maxBVHDepth) maxBVHDepth = o.maxDepthBVH;
        // GetFastBV (:454-479)
        V3 lmin = loaded[o.meshId].localMin, lmax = loaded[o.meshId].localMax;
        V3 corners[8] = {lmin, {lmin.x, lmin.y, lmax.z}, {lmin.x, lmax.y, lmin.z}, {lmin.x, lmax.y, lmax.z},
                         {lmax.x, lmin.y, lmin.z}, {lmax.x, lmin.y, lmax.z}, {lmax.x, lmax.y, lmin.z}, lmax};
        V3 wmin = {FLT_MAX, FLT_MAX, FLT_MAX}, wmax = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (auto& c : corners) { V3 w = mulPoint(l2w, c); wmin = vmin(wmin, w); wmax = vmax(wmax, w); }
        d.min = toR3(wmin);
        d.max = toR3(wmax);
        d.areaApprox = boxAreaStored(wmin, wmax);
        totArea += d.areaApprox;  // the reference adds these from a Parallel.For (:618-636); order fixed here
        objs[i] = d;
    }
    std::vector<Comp> ocomps(nObj);
    for (int i = 0; i < nObj; i++) { ocomps[i].mn = toV3(objs[i].min); ocomps[i].mx = toV3(objs[i].max); ocomps[i].payload = i; }
    Builder sb;
    sb.splitResolution = s->splitResolution;
    sb.build(ocomps, s->maxDepthSceneBVH);
    int sceneBVHStartIndex = (int)s->BVs.size();
    for (BVNode bv : sb.nodes) {
        if (bv.childIndexA != -1) bv.childIndexA += sceneBVHStartIndex;
        if (bv.childIndexB != -1) bv.childIndexB += sceneBVHStartIndex;
        s->BVs.push_back(toRenderBV(bv));
        s->componentDatas.push_back({bv.componentStart, bv.totComponents});
    }
    // SceneObjectDatas is reordered in place by the scene-BVH partition (:644)
    s->sceneObjects.resize(nObj);
    for (int i = 0; i < nObj; i++) s->sceneObjects[i] = objs[ocomps[i].payload];

    // ---- light list (:694-712) ----
    s->lightObjects.clear();
    int emitting = 0;
    for (const auto& o : s->objects)
        if (s->materialInputs[o.materialIndex].brightness != 0.0f) emitting++;
    for (const auto& so : s->sceneObjects) {
        float brightness = s->materialInputs[so.materialIndex].brightness;
        if (brightness != 0.0f) {
            rt_light_object l;
            memset(&l, 0, sizeof l);
            l.localToWorldMatrix = so.localToWorldMatrix;
            l.areaApprox = so.areaApprox;
            l.brightness = brightness;
            l.triStart = s->componentDatas[so.bvStartIndex].x;
            l.totTris = s->componentDatas[so.bvStartIndex].y;
            s->lightObjects.push_back(l);
        }
    }
    s->maxBVHDepth = maxBVHDepth;
    s->emittingObjectsNum = emitting;
    s->sceneBVHStartIndex = sceneBVHStartIndex;
    s->lastSceneBVHLength = (int)sb.nodes.size();
    s->totArea = (float)totArea;
    s->built = true;
}

}  // namespace

extern "C" {

rth_scene rth_scene_create(int splitResolution, int maxDepthSceneBVH) {
    if (splitResolution < 1) return nullptr;
    rth_scene s = new rth_scene_s;
    s->splitResolution = splitResolution;
    s->maxDepthSceneBVH = maxDepthSceneBVH;
    return s;
}
void rth_scene_free(rth_scene s) { delete s; }
int rth_scene_add_material(rth_scene s, const rth_material_input* m) {
    if (!s || !m) return -1;
    s->materialInputs.push_back(*m);
    s->materialTextures.emplace_back();
    return (int)s->materialInputs.size() - 1;
}
int rth_scene_add_material_textured(rth_scene s, const rth_material_input* m, const rth_texture_input tex[5]) {
    if (!s || !m || !tex) return -1;
    std::array<rth_scene_s::Tex, 5> t;
    for (int k = 0; k < 5; k++) {
        if (!tex[k].pixels) continue;
        if (tex[k].width < 1 || tex[k].height < 1 || tex[k].width > 8192 || tex[k].height > 8192) return -1;
        t[(size_t)k].w = tex[k].width; t[(size_t)k].h = tex[k].height;
        t[(size_t)k].px.assign(tex[k].pixels, tex[k].pixels + (size_t)tex[k].width * tex[k].height * 4);
    }
    s->materialInputs.push_back(*m);
    s->materialTextures.push_back(std::move(t));
    return (int)s->materialInputs.size() - 1;
}
int rth_scene_add_mesh(rth_scene s, rth_mesh m) {
    if (!s || !m) return -1;
    s->meshes.push_back(*m);
    return (int)s->meshes.size() - 1;
}
int rth_scene_add_object(rth_scene s, int meshId, const float position[3], const float eulerDeg[3], const float scale[3],
                         int materialIndex, int maxDepthBVH) {
    if (!s || meshId < 0 || meshId >= (int)s->meshes.size() || materialIndex < 0 ||
        materialIndex >= (int)s->materialInputs.size())
        return -1;
    ObjectDesc o;
    o.meshId = meshId;
    memcpy(o.pos, position, 12);
    memcpy(o.rot, eulerDeg, 12);
    memcpy(o.scale, scale, 12);
    o.materialIndex = materialIndex;
    o.maxDepthBVH = maxDepthBVH;
    s->objects.push_back(o);
    return (int)s->objects.size() - 1;
}

void rth_world_to_local(const float position[3], const float eulerDeg[3], const float scale[3], float outW2L[16],
                        float outL2W[16]) {
    M4 w2l = createWorldToLocal(position, eulerDeg, scale);
    M4 l2w = inverse(w2l);
    memcpy(outW2L, w2l.m, 64);
    memcpy(outL2W, l2w.m, 64);
}

int rth_scene_build(rth_scene s, int threads) {
    if (!s || s->objects.empty()) return -1;
    if (threads <= 0) threads = omp_get_max_threads();
    const int nMesh = (int)s->meshes.size();

    if (!rth_scene_make_materials(s)) return -1;

    // ---- LoadSceneObjects (:556-607): meshes get their BVH the first time an object uses them ----
    std::vector<int> firstUseOrder;          // mesh ids in order of first use
    std::vector<int> meshMaxDepth(nMesh, 0); // the depth of the first object that uses the mesh
    std::vector<char> used(nMesh, 0);
    for (const auto& o : s->objects)
        if (!used[o.meshId]) { used[o.meshId] = 1; firstUseOrder.push_back(o.meshId); meshMaxDepth[o.meshId] = o.maxDepthBVH; }

    struct MeshBuild { std::vector<BVNode> nodes; std::vector<Comp> comps; std::vector<float> area; };
    std::vector<MeshBuild> builds(nMesh);
    const int nUse = (int)firstUseOrder.size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int u = 0; u < nUse; u++) {
        int id = firstUseOrder[u];
        const rth_mesh_s& mesh = s->meshes[id];
        MeshBuild& mb = builds[id];
        size_t nt = mesh.tris.size() / 3;
        mb.comps.resize(nt);
        mb.area.resize(nt);
        for (size_t t = 0; t < nt; t++) {  // Triangle.Initialise (Triangle.cs:17-46)
            V3 v0 = toV3(mesh.verts[mesh.tris[3 * t]].pos), v1 = toV3(mesh.verts[mesh.tris[3 * t + 1]].pos),
               v2 = toV3(mesh.verts[mesh.tris[3 * t + 2]].pos);
            mb.comps[t].mn = vmin(vmin(v0, v1), v2);
            mb.comps[t].mx = vmax(vmax(v0, v1), v2);
            mb.comps[t].payload = (int32_t)t;
            V3 ab = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z}, ac = {v2.x - v0.x, v2.y - v0.y, v2.z - v0.z};
            V3 cr = {ab.y * ac.z - ab.z * ac.y, ab.z * ac.x - ab.x * ac.z, ab.x * ac.y - ab.y * ac.x};
            mb.area[t] = sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z) * 0.5f;  // Func.GetTriArea (Resources.cs:294-301)
        }
        Builder b;
        b.splitResolution = s->splitResolution;
        b.build(mb.comps, meshMaxDepth[id]);
        mb.nodes.swap(b.nodes);
    }

    // ---- concatenate in first-use order: ConstructBVH (:303-347) ----
    s->BVs.clear(); s->componentDatas.clear(); s->vertices.clear(); s->triangles.clear();
    std::vector<LoadedMesh>& loaded = s->loaded;
    loaded.assign(nMesh, LoadedMesh());
    for (int id : firstUseOrder) {
        const rth_mesh_s& mesh = s->meshes[id];
        MeshBuild& mb = builds[id];
        int loadedBVsLength = (int)s->BVs.size();
        int loadedTrianglesLength = (int)s->triangles.size();
        int baseVertexIndex = (int)s->vertices.size();
        LoadedMesh& lm = loaded[id];
        lm.bvStartIndex = loadedBVsLength;
        lm.componentStartIndex = loadedTrianglesLength;
        lm.localMin = mb.nodes[0].mn;  // GetMin/GetMax over the mesh's triangles (:591)
        lm.localMax = mb.nodes[0].mx;
        lm.built = true;
        for (BVNode bv : mb.nodes) {
            if (bv.childIndexA != -1) bv.childIndexA += loadedBVsLength;
            if (bv.childIndexB != -1) bv.childIndexB += loadedBVsLength;
            bv.componentStart += loadedTrianglesLength;
            s->BVs.push_back(toRenderBV(bv));
            s->componentDatas.push_back({bv.componentStart, bv.totComponents});
        }
        bool containsUVs = true;  // .rtmesh always carries uvs ((-1,-1) where the OBJ had none)
        (void)containsUVs;
        for (const auto& v : mesh.verts) s->vertices.push_back(v);
        for (const Comp& c : mb.comps) {  // RenderTriangle pack (:715-729), BVH-sorted order
            rt_triangle t;
            memset(&t, 0, sizeof t);
            t.vertex0Index = mesh.tris[3 * (size_t)c.payload] + baseVertexIndex;
            t.vertex1Index = mesh.tris[3 * (size_t)c.payload + 1] + baseVertexIndex;
            t.vertex2Index = mesh.tris[3 * (size_t)c.payload + 2] + baseVertexIndex;
            t.area = mb.area[(size_t)c.payload];
            s->triangles.push_back(t);
        }
        mb = MeshBuild();
    }

    constructSceneLevel(s);
    return 0;
}

// A GameObject moved (its SceneObjectSettings.LateUpdate would set Main.DoReloadData, SceneObjectSettings.cs:27-36).
int rth_scene_set_object_transform(rth_scene s, int objectIndex, const float position[3], const float eulerDeg[3], const float scale[3]) {
    if (!s || objectIndex < 0 || objectIndex >= (int)s->objects.size() || !position || !eulerDeg || !scale) return -1;
    ObjectDesc& o = s->objects[(size_t)objectIndex];
    memcpy(o.pos, position, 12);
    memcpy(o.rot, eulerDeg, 12);
    memcpy(o.scale, scale, 12);
    return 0;
}

// ConstructScene called again on a built scene (Main.SetData on a reload, Main.cs:278): every mesh is found among the
// loaded ones (ObjectHelper.cs:582-584), so no mesh BVH is rebuilt and the vertex and triangle arrays stay as they are;
// the per-object data, the scene BVH and the light list are made anew.
int rth_scene_rebuild(rth_scene s, int keepOldSceneBVH) {
    if (!s || !s->built) return -1;
    if (s->objects.empty()) return 0;  // filled from a scene cache: LoadExistingFile reads the same files again (:672-677)
    for (const auto& o : s->objects)
        if (o.meshId < 0 || o.meshId >= (int)s->loaded.size() || !s->loaded[(size_t)o.meshId].built) return -1;
    if (!keepOldSceneBVH) {  // Utils.RemoveFromEndOfArray(ref loadedBoundingVolumes, lastSceneBVHLength), MeshHelper.cs:371-372
        s->BVs.resize(s->BVs.size() - (size_t)s->lastSceneBVHLength);
        s->componentDatas.resize(s->componentDatas.size() - (size_t)s->lastSceneBVHLength);
    }
    constructSceneLevel(s);
    return 0;
}

}  // extern "C"

// Internal door for scene_cache.cpp: the arrays ConstructScene returns, for writing them out or filling them in.
SceneArraysRef rth_scene_arrays_ref(rth_scene s) {
    SceneArraysRef r = {&s->BVs, &s->componentDatas, &s->vertices, &s->triangles, &s->sceneObjects, &s->lightObjects,
                        &s->maxBVHDepth, &s->emittingObjectsNum, &s->sceneBVHStartIndex, &s->totArea, &s->built};
    return r;
}
// ConstructTextureAtlas (ObjectHelper.cs:370-452): packs the materials' textures into one atlas and fills one Material2
// per material input.  Returns false when a material refers to a rectangle that was never packed.
bool rth_scene_make_materials(rth_scene s) {
    // the list that is packed: col, bump, roughness, normals of every material, in that order (:373-379)
    struct Item { const rth_scene_s::Tex* t; int x = 0, y = 0; };
    std::vector<Item> items;
    for (const auto& mt : s->materialTextures)
        for (int k : {RTH_TEX_COL, RTH_TEX_BUMP, RTH_TEX_ROUGHNESS, RTH_TEX_NORMALS})
            if (mt[(size_t)k].w) items.push_back({&mt[(size_t)k]});
    s->atlasW = s->atlasH = 1;
    s->atlas.assign(4, 0);
    if (!items.empty()) {
        // shelf packing with 1 texel of padding, into the smallest power-of-two square that takes everything
        const int pad = 1;
        for (int size = 2;; size *= 2) {
            int x = 0, y = 0, shelf = 0;
            bool fits = true;
            for (auto& it : items) {
                if (it.t->w + pad > size) { fits = false; break; }
                if (x + it.t->w + pad > size) { x = 0; y += shelf; shelf = 0; }
                if (y + it.t->h + pad > size) { fits = false; break; }
                it.x = x; it.y = y;
                x += it.t->w + pad;
                if (it.t->h + pad > shelf) shelf = it.t->h + pad;
            }
            if (fits) { s->atlasW = s->atlasH = size; break; }
            if (size >= 16384) return false;
        }
        s->atlas.assign((size_t)s->atlasW * s->atlasH * 4, 0);
        for (const auto& it : items)
            for (int r = 0; r < it.t->h; r++)
                memcpy(&s->atlas[((size_t)(it.y + r) * s->atlasW + it.x) * 4], &it.t->px[(size_t)r * it.t->w * 4], (size_t)it.t->w * 4);
    }
    // Rect[] of PackTextures is normalised; GetTexLoc / GetTexDims scale back and truncate (:386-387)
    auto loc = [&](const Item& it) { return rt_int2{(int)(((float)it.x / (float)s->atlasW) * (float)s->atlasW), (int)(((float)it.y / (float)s->atlasH) * (float)s->atlasH)}; };
    auto dims = [&](const Item& it) { return rt_int2{(int)(((float)it.t->w / (float)s->atlasW) * (float)s->atlasW), (int)(((float)it.t->h / (float)s->atlasH) * (float)s->atlasH)}; };
    size_t rectIndex = 0;
    bool ok = true;
    auto take = [&](rt_int2& outLoc, rt_int2& outDims) {
        if (rectIndex >= items.size()) { ok = false; return; }
        outLoc = loc(items[rectIndex]);
        outDims = dims(items[rectIndex]);
        rectIndex++;
    };
    s->materials.clear();
    for (size_t i = 0; i < s->materialInputs.size(); i++) {
        const auto& mi = s->materialInputs[i];
        const auto& mt = s->materialTextures[i];
        rt_material m;
        memset(&m, 0, sizeof m);
        m.brightness = mi.brightness;
        if (mt[RTH_TEX_COL].w) { take(m.colTexLoc, m.colTexDims); m.col = {-1, -1, -1}; } else m.col = {mi.col[0], mi.col[1], mi.col[2]};
        if (mt[RTH_TEX_BUMP].w) { take(m.bumpTexLoc, m.bumpTexDims); m.bump = -1; } else m.bump = mi.bump;
        if (mt[RTH_TEX_ROUGHNESS].w) { take(m.roughnessTexLoc, m.roughnessTexDims); m.roughness = -1; } else m.roughness = mi.roughness;
        // consumes a rectangle although metallicity textures were never packed (:430-437)
        if (mt[RTH_TEX_METALLICITY].w) { take(m.metallicityTexLoc, m.metallicityTexDims); m.metallicity = -1; } else m.metallicity = mi.metallicity;
        if (mt[RTH_TEX_NORMALS].w) take(m.normalsTexLoc, m.normalsTexDims); else m.normalsTexLoc = {-1, -1};
        s->materials.push_back(m);
    }
    return ok;
}

extern "C" {

int rth_scene_get(rth_scene s, rth_scene_data* out) {
    if (!s || !out || !s->built) return -1;
    out->BVs = s->BVs.data(); out->nBVs = (int64_t)s->BVs.size();
    out->Vertices = s->vertices.data(); out->nVertices = (int64_t)s->vertices.size();
    out->Triangles = s->triangles.data(); out->nTriangles = (int64_t)s->triangles.size();
    out->SceneObjects = s->sceneObjects.data(); out->nSceneObjects = (int64_t)s->sceneObjects.size();
    out->LightObjects = s->lightObjects.data(); out->nLightObjects = (int64_t)s->lightObjects.size();
    out->Materials = s->materials.data(); out->nMaterials = (int64_t)s->materials.size();
    out->ComponentDatas = s->componentDatas.data(); out->nComponentDatas = (int64_t)s->componentDatas.size();
    out->maxBVHDepth = s->maxBVHDepth;
    out->emittingObjectsNum = s->emittingObjectsNum;
    out->sceneBVHStartIndex = s->sceneBVHStartIndex;
    out->totArea = s->totArea;
    out->atlas = s->atlas.data(); out->atlasWidth = s->atlasW; out->atlasHeight = s->atlasH;
    return 0;
}

}  // extern "C"
