// rt_oracle_ris.cpp — CPU restatement of the RIS project's compute kernels (BVH_RIS_DL).
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle_ris.h).  PARITY UNPINNED against the reference's own execution.
// Every function cites the reference lines it restates; "RI:" = BVH_RIS_DL/Assets/Scripts/ComputeShaders/
// RIS_BVH_RayTracer.compute, "RR:" = .../RendererResources.hlsl, "RD:" = .../RendererDataTypes.hlsl.
//
// Conventions beyond those of orc_math.h: lerp(x,y,s) = x + s*(y-x); reflect(i,n) = i - (2*n)*dot(i,n);
// the uninitialised `combinedTraceInfo.incomingLight` (RI:512,551) starts at zero.
#include "rt_oracle_ris.h"
#include "orc_math.h"

#include <omp.h>

namespace {

using namespace orcm;

// RR:85-92: the RIS variant compares against r*(a+b), the ReSTIR one against r*b.
inline bool weightedRandRIS(float a, float b, uint32_t& state) {
    float randNorm = randNormalized(state);
    float totalWeight = a + b;
    float relRand = randNorm * totalWeight;
    return relRand < a;
}
inline f3 lerp3(f3 x, f3 y, float s) { return x + (y - x) * s; }
inline f3 reflect3(f3 i, f3 n) { return i - (n * 2.0f) * dot(i, n); }
inline bool anyZero(f3 v) { return v.x == 0 || v.y == 0 || v.z == 0; }  // `all(v) == 0`, RI:463,525

struct Ray { f3 pos, dir; };                                                    // RD:50-54
struct HitInfo { float dst; f3 hitPoint; f2 uv; f3 normal; int materialIndex; };  // RD:55-62
struct DebugData { int triChecks, bvChecks; };
struct RayStats {
    uint32_t visitHash = 2166136261u;
    int triIndex = -1, objIndex = -1, curTri = -1;
    uint64_t pops = 0, bvChecks = 0, objVisits = 0, triIters = 0, triChecks = 0;
    void visitInternal(uint32_t node) { visitHash = (visitHash ^ (0x80000000u | node)) * 16777619u; }
    void visitLeaf(int a, int b) {
        visitHash = (visitHash ^ (uint32_t)a) * 16777619u;
        visitHash = (visitHash ^ (uint32_t)b) * 16777619u;
    }
};
inline HitInfo InitHitInfo() {  // RD:107-117
    HitInfo h;
    h.dst = INF;
    h.hitPoint = F3(-1, -1, -1);
    h.uv = {0, 0};
    h.normal = F3(0, 0, 0);
    h.materialIndex = 0;
    return h;
}
const int STACK_SLACK = 64;  // the reference's stacks overflow at 33 entries; the slack keeps the oracle defined

struct Ctx {
    const orc_ris_scene& s;
    const orc_ris_params& p;
    orc_ris_frame& f;
    int W, H;
    Ctx(const orc_ris_scene& s_, const orc_ris_params& p_, orc_ris_frame& f_) : s(s_), p(p_), f(f_) {
        W = p.Resolution[0];
        H = p.Resolution[1];
    }

    // Store to a render target; 8-bit UNORM targets (Main.cs:324-364: `new RenderTexture(w, h, 24)`) convert as D3D/Metal
    // do (NaN -> 0, clamp, scale by 255, add 0.5, truncate) and keep the float the next load would return.
    static float toUnorm8(float x) { return (float)(int)(saturate(x) * 255.0f + 0.5f) / 255.0f; }
    void storeRT(rt_float4* tex, int64_t px, rt_float4 v) const {
        if (p.unorm8Targets) v = {toUnorm8(v.x), toUnorm8(v.y), toUnorm8(v.z), toUnorm8(v.w)};
        tex[px] = v;
    }

    // RI:86-109
    static bool RayTriangleIntersect(const Ray& ray, const rt_ris_tri& tri, float& dstOut, f2& uvOut) {
        f3 edgeAB = F3(tri.vB) - F3(tri.vA);
        f3 edgeAC = F3(tri.vC) - F3(tri.vA);
        f3 normalVector = cross(edgeAB, edgeAC);
        f3 ao = ray.pos - F3(tri.vA);
        f3 dao = cross(ao, ray.dir);
        float determinant = -dot(ray.dir, normalVector);
        float invDet = 1 / determinant;
        float dst = dot(ao, normalVector) * invDet;
        float u = dot(edgeAC, dao) * invDet;
        float v = -dot(edgeAB, dao) * invDet;
        float w = 1 - u - v;
        bool didHit = determinant >= 1E-8f && dst >= 0 && u >= 0 && v >= 0 && w >= 0;
        if (didHit) {
            uvOut.x = (tri.uvA.x * w + tri.uvB.x * u) + tri.uvC.x * v;
            uvOut.y = (tri.uvA.y * w + tri.uvB.y * u) + tri.uvC.y * v;
        } else {
            uvOut = {0, 0};
        }
        dstOut = dst;
        return didHit;
    }
    // RI:111-124: divides by the direction (the ReSTIR project multiplies by a stored reciprocal).
    static float RayBVIntersect(const Ray& ray, const rt_bv& bv) {
        f3 a = F3(bv.min) - ray.pos, b = F3(bv.max) - ray.pos;
        f3 tMin = {a.x / ray.dir.x, a.y / ray.dir.y, a.z / ray.dir.z};
        f3 tMax = {b.x / ray.dir.x, b.y / ray.dir.y, b.z / ray.dir.z};
        f3 t1 = min3(tMin, tMax);
        f3 t2 = max3(tMin, tMax);
        float tNear = fmaxf_(fmaxf_(t1.x, t1.y), t1.z);
        float tFar = fminf_(fminf_(t2.x, t2.y), t2.z);
        bool didHit = tFar >= tNear && tFar > 0;
        return didHit ? (tNear > 0 ? tNear : 0) : INF;
    }
    // RI:126-176: no back-face pre-cull; the hit keeps the triangle's stored world normal.
    void TraverseTrisBVH(HitInfo& hitInfo, DebugData& debugData, const Ray& ray, int bvStartIndex, RayStats& st) const {
        int bvStack[MAX_BVH_DEPTH + 1 + STACK_SLACK];
        int stackPointer = 0;
        bvStack[stackPointer++] = bvStartIndex;
        while (stackPointer > 0) {
            int node = bvStack[--stackPointer];
            rt_bv bv = ld(s.BVs, s.nBVs, node);
            st.pops++;
            bool isLeaf = bv.indexA >= 0;
            if (isLeaf) st.visitLeaf(bv.indexA, bv.indexB); else st.visitInternal((uint32_t)node);
            bv.indexA = bv.indexA < 0 ? -bv.indexA : bv.indexA;
            bv.indexB = bv.indexB < 0 ? -bv.indexB : bv.indexB;
            if (isLeaf) {
                for (int triIndex = bv.indexA; triIndex < bv.indexA + bv.indexB; triIndex++) {
                    rt_ris_tri tri = ld(s.Tris, s.nTris, triIndex);
                    float dst; f2 uv;
                    bool didHit = RayTriangleIntersect(ray, tri, dst, uv);
                    debugData.triChecks++;
                    st.triIters++;
                    st.triChecks++;
                    if (didHit && dst < hitInfo.dst) {
                        hitInfo.dst = dst;
                        hitInfo.uv = uv;
                        hitInfo.normal = F3(tri.worldNormal);
                        st.curTri = triIndex;
                    }
                }
            } else {
                rt_bv bvA = ld(s.BVs, s.nBVs, bv.indexA);
                float dstA = RayBVIntersect(ray, bvA); debugData.bvChecks++;
                rt_bv bvB = ld(s.BVs, s.nBVs, bv.indexB);
                float dstB = RayBVIntersect(ray, bvB); debugData.bvChecks++;
                st.bvChecks += 2;
                if (stackPointer + 2 > (int)MAX_BVH_DEPTH + 1 + STACK_SLACK) continue;
                if (dstA < dstB) {
                    if (dstB < hitInfo.dst) bvStack[stackPointer++] = bv.indexB;
                    if (dstA < hitInfo.dst) bvStack[stackPointer++] = bv.indexA;
                } else {
                    if (dstA < hitInfo.dst) bvStack[stackPointer++] = bv.indexA;
                    if (dstB < hitInfo.dst) bvStack[stackPointer++] = bv.indexB;
                }
            }
        }
    }
    // RI:178-238
    void TraverseSceneBVH(HitInfo& hitInfo, DebugData& debugData, const Ray& ray, RayStats& st) const {
        int bvStack[MAX_BVH_DEPTH + 1 + STACK_SLACK];
        int stackPointer = 0;
        bvStack[stackPointer++] = p.SceneBVHStartIndex;
        while (stackPointer > 0) {
            int node = bvStack[--stackPointer];
            rt_bv bv = ld(s.BVs, s.nBVs, node);
            st.pops++;
            bool isLeaf = bv.indexA >= 0;
            if (isLeaf) st.visitLeaf(bv.indexA, bv.indexB); else st.visitInternal((uint32_t)node);
            bv.indexA = bv.indexA < 0 ? -bv.indexA : bv.indexA;
            bv.indexB = bv.indexB < 0 ? -bv.indexB : bv.indexB;
            if (isLeaf) {
                for (int sceneObjectIndex = bv.indexA; sceneObjectIndex < bv.indexA + bv.indexB; sceneObjectIndex++) {
                    rt_scene_object sceneObject = ld(s.SceneObjects, s.nSceneObjects, sceneObjectIndex);
                    st.objVisits++;
                    Ray localRay;
                    localRay.pos = mul44(sceneObject.worldToLocalMatrix, ray.pos, 1);
                    localRay.dir = mul44(sceneObject.worldToLocalMatrix, ray.dir, 0);
                    HitInfo localHitInfo = InitHitInfo();
                    st.curTri = -1;
                    TraverseTrisBVH(localHitInfo, debugData, localRay, sceneObject.bvStartIndex, st);
                    if (localHitInfo.dst < hitInfo.dst) {
                        hitInfo.dst = localHitInfo.dst;
                        hitInfo.normal = localHitInfo.normal;
                        hitInfo.hitPoint = ray.pos + ray.dir * hitInfo.dst;
                        hitInfo.uv = localHitInfo.uv;
                        hitInfo.materialIndex = sceneObject.materialIndex;
                        st.triIndex = st.curTri;
                        st.objIndex = sceneObjectIndex;
                    }
                }
            } else {
                rt_bv bvA = ld(s.BVs, s.nBVs, bv.indexA);
                float dstA = RayBVIntersect(ray, bvA); debugData.bvChecks++;
                rt_bv bvB = ld(s.BVs, s.nBVs, bv.indexB);
                float dstB = RayBVIntersect(ray, bvB); debugData.bvChecks++;
                st.bvChecks += 2;
                if (stackPointer + 2 > (int)MAX_BVH_DEPTH + 1 + STACK_SLACK) continue;
                if (dstA < dstB) {
                    if (dstB < hitInfo.dst) bvStack[stackPointer++] = bv.indexB;
                    if (dstA < hitInfo.dst) bvStack[stackPointer++] = bv.indexA;
                } else {
                    if (dstA < hitInfo.dst) bvStack[stackPointer++] = bv.indexA;
                    if (dstB < hitInfo.dst) bvStack[stackPointer++] = bv.indexB;
                }
            }
        }
    }
    HitInfo GetRayCollision(const Ray& ray, DebugData& debugData, RayStats& st) const {  // RI:240-247
        HitInfo hitInfo = InitHitInfo();
        TraverseSceneBVH(hitInfo, debugData, ray, st);
        return hitInfo;
    }

    static float GetTriArea(const rt_ris_tri& tri) {  // RR:117-124
        f3 ab = F3(tri.vB) - F3(tri.vA);
        f3 ac = F3(tri.vC) - F3(tri.vA);
        return length(cross(ab, ac)) * 0.5f;
    }
    // RR:126-143
    static f3 GetRandWorldPointTri(const rt_ris_tri& tri, const rt_float4x4& localToWorldMatrix, uint32_t& state) {
        float r1 = randNormalized(state);
        float r2 = randNormalized(state);
        if (r1 + r2 > 1.0f) {
            r1 = 1.0f - r1;
            r2 = 1.0f - r2;
        }
        f3 localPoint = F3(tri.vA) * (1.0f - r1 - r2) + F3(tri.vB) * r1 + F3(tri.vC) * r2;
        return mul44(localToWorldMatrix, localPoint, 1.0f);
    }
    // RI:273-293
    int SceneObjectReservoirSample(uint32_t& state) const {
        int firstIndex = randIntSpan(0, p.EmittingObjectsNum, state);
        float totWeights = ld(s.LightObjects, s.nLightObjects, firstIndex).areaApprox;
        int chosenIndex = firstIndex;
        for (int i = 0; i < p.SceneObjectReservoirTestsNum; i++) {
            int candidateIndex = randIntSpan(0, p.EmittingObjectsNum, state);
            float candidateWeight = ld(s.LightObjects, s.nLightObjects, candidateIndex).areaApprox;
            if (weightedRandRIS(candidateWeight, totWeights, state)) chosenIndex = candidateIndex;
            totWeights += candidateWeight;
        }
        return chosenIndex;
    }
    // RI:295-315
    int TriReservoirSample(int startTriIndex, int endTriIndex, uint32_t& state) const {
        int firstIndex = randIntSpan(startTriIndex, endTriIndex, state);
        float totWeights = GetTriArea(ld(s.Tris, s.nTris, firstIndex));
        int chosenIndex = firstIndex;
        for (int i = 0; i < p.TriReservoirTestsNum; i++) {
            int candidateIndex = randIntSpan(startTriIndex, endTriIndex, state);
            float candidateWeight = GetTriArea(ld(s.Tris, s.nTris, candidateIndex));
            if (weightedRandRIS(candidateWeight, totWeights, state)) chosenIndex = candidateIndex;
            totWeights += candidateWeight;
        }
        return chosenIndex;
    }
    void storeCandidate(int index, rt_float4 v) const {
        if (index >= 0 && (int64_t)index < f.nCandidates) f.Candidates[index] = v;
    }
    rt_float4 loadCandidate(int index) const { return ld(f.Candidates, f.nCandidates, index); }
    // RI:317-359
    void GenerateCandidates2(f3 hitNormal, f3 pos, int rayIndex, uint32_t& state) const {
        for (int i = 0; i < p.SceneObjectCandidatesNum; i++) {
            int lightObjectIndex = SceneObjectReservoirSample(state);
            rt_light_object lightObject = ld(s.LightObjects, s.nLightObjects, lightObjectIndex);
            int startTriIndex = lightObject.triStart;
            int endTriIndex = lightObject.triStart + lightObject.totTris;
            for (int j = 0; j < p.TriCandidatesNum; j++) {
                int triIndex = TriReservoirSample(startTriIndex, endTriIndex, state);
                rt_ris_tri tri = ld(s.Tris, s.nTris, triIndex);
                f3 worldPoint = GetRandWorldPointTri(tri, lightObject.localToWorldMatrix, state);
                f3 dst = worldPoint - pos;
                f3 dir = normalize(dst);
                int candidateIndex = rayIndex * p.TotCandidatesNum + i * p.TriCandidatesNum + j;
                if (dot(hitNormal, dir) < 0) {
                    storeCandidate(candidateIndex, {0, 0, 0, 0});
                    continue;
                }
                float hitAngleDot = fmaxf_(dot(F3(tri.worldNormal), dir), 0);
                float BRDF = 1.0f;
                float weight = lightObject.brightness * BRDF * hitAngleDot / dot2(dst);
                storeCandidate(candidateIndex, {dir.x, dir.y, dir.z, weight});
            }
        }
    }

    // ---- kernel GenerateCandidates, RI:384-412 (InitialTrace RI:361-376) ----
    void GenerateCandidates(uint32_t idx, uint32_t idy, orc_counters& cnt) const {
        DebugData debugData = {0, 0};
        uint32_t state = idy * (uint32_t)p.Resolution[0] + idx + (uint32_t)p.FrameRand;
        float focalPosX = ((float)idx / (float)p.Resolution[0] - 0.5f) * 2.0f * p.ViewSpaceDims[0];
        float focalPosY = ((float)idy / (float)p.Resolution[1] - 0.5f) * 2.0f * p.ViewSpaceDims[1];
        f3 focalPos = F3(focalPosX, focalPosY, 1) * p.FocalPlaneFactor;
        f3 cam = F3(p.WorldSpaceCameraPos[0], p.WorldSpaceCameraPos[1], p.WorldSpaceCameraPos[2]);
        f3 worldFocalPos = mul33(p.CameraTransform, focalPos) + cam;
        uint32_t rayIndex = ((uint32_t)p.Resolution[0] * idy + idx) * (uint32_t)p.RaysPerPixel;
        Ray ray;
        ray.pos = cam;
        ray.dir = normalize(worldFocalPos - ray.pos);

        RayStats st;
        HitInfo hitInfo = GetRayCollision(ray, debugData, st);
        bool didHit = hitInfo.dst != INF;
        if (didHit) GenerateCandidates2(hitInfo.normal, hitInfo.hitPoint, (int)rayIndex, state);
        if ((int64_t)rayIndex < f.nHitInfos) {
            rt_ris_hit_info out;
            out.dst = hitInfo.dst; out.hitPoint = R3(hitInfo.hitPoint); out.uv = {hitInfo.uv.x, hitInfo.uv.y};
            out.normal = R3(hitInfo.normal); out.materialIndex = hitInfo.materialIndex;
            f.HitInfos[rayIndex] = out;
        }
        int64_t px = (int64_t)idy * W + idx;
        f3 firstHit = didHit ? hitInfo.hitPoint : F3(0, 0, 0);
        storeRT(f.RayHitPoints, px, {firstHit.x, firstHit.y, firstHit.z, 0});
        bool badMaxBVHDepth = p.MaxBVHDepth >= (int)MAX_BVH_DEPTH;  // `>=` here, `>` in the ReSTIR project
        storeRT(f.DebugOverlay, px, {debugData.triChecks / (float)p.DebugDataMaxValues[0],
                                     debugData.bvChecks / (float)p.DebugDataMaxValues[1], badMaxBVHDepth ? 1.0f : 0.0f, 0});
        if (f.records) {
            orc_trace_record& r = f.records[px];
            r.triIndex = st.triIndex; r.objIndex = st.objIndex; r.visitHash = st.visitHash;
            r.lightObj = -1; r.lightTri = -1; r.shadowVisitHash = 0; r.shadowTri = -1; r.bounceVisitHash = 0;
        }
        cnt.rays += 1; cnt.pops += st.pops; cnt.bvChecks += st.bvChecks; cnt.objVisits += st.objVisits;
        cnt.triIters += st.triIters; cnt.triChecks += st.triChecks; cnt.activePixels++;
    }

    // ---- kernel TraceRays ----
    // RI:416-453.  The loop re-reads Candidates[firstIndex] (RI:425), not the index it just drew.
    rt_float4 RayDirFromCandidates(uint32_t rayIndex, uint32_t& state) const {
        int startIndex = (int)(rayIndex * (uint32_t)p.TotCandidatesNum);
        int endIndex = startIndex + p.TotCandidatesNum;
        int firstIndex = randIntSpan(startIndex, endIndex, state);
        float firstWeight = loadCandidate(firstIndex).w;
        int chosenIndex = firstIndex;
        float totWeights = firstWeight;
        for (int i = 0; i < p.CandidateReservoirTestsNum; i++) {
            int candidateIndex = randIntSpan(startIndex, endIndex, state);
            rt_float4 candidate = loadCandidate(firstIndex);
            if (candidate.w == 0) continue;
            float candidateWeight = candidate.w;
            if (weightedRandRIS(candidateWeight, totWeights, state)) chosenIndex = candidateIndex;
            totWeights += candidateWeight;
        }
        rt_float4 chosenCandidate = loadCandidate(chosenIndex);
        float correctionWeight = totWeights / ((float)p.CandidateReservoirTestsNum * chosenCandidate.w);
        return {chosenCandidate.x, chosenCandidate.y, chosenCandidate.z, correctionWeight};
    }
    // RI:455-501
    f3 Trace(Ray ray, HitInfo hitInfo, DebugData& debugData, uint32_t& state, RayStats& st, uint64_t& rays) const {
        f3 rayColor = F3(1, 1, 1), incomingLight = F3(0, 0, 0);
        for (int bounceCount = 0; bounceCount <= p.MaxBounceCount; bounceCount++) {
            if (hitInfo.dst == INF || anyZero(ray.dir)) break;
            rt_ris_material material = ld(s.Materials, s.nMaterials, hitInfo.materialIndex);
            ray.pos = hitInfo.hitPoint - hitInfo.normal * 0.001f;
            bool isSpecularBounce = false;
            if (bounceCount != 0) {
                isSpecularBounce = p.ScatterProbability < randNormalized(state);
                f3 scatterDir = normalize(hitInfo.normal + randPointOnUnitSphere(state));
                f3 specularDir = reflect3(ray.dir, hitInfo.normal);
                ray.dir = normalize(lerp3(scatterDir, specularDir, material.smoothness * (isSpecularBounce ? 1.0f : 0.0f)));
                if (dot(hitInfo.normal, ray.dir) < 0) ray.dir = -ray.dir;
            }
            f3 sampleCol = F3(material.color);
            f3 emittedLight = sampleCol * material.brightness;
            incomingLight = incomingLight + emittedLight * rayColor / sqr(hitInfo.dst);
            rayColor = rayColor * lerp3(F3(material.color), F3(material.specularColor), isSpecularBounce ? 1.0f : 0.0f);
            if (bounceCount != p.MaxBounceCount) {
                RayStats one;
                hitInfo = GetRayCollision(ray, debugData, one);
                rays++;
                st.pops += one.pops; st.bvChecks += one.bvChecks; st.objVisits += one.objVisits;
                st.triIters += one.triIters; st.triChecks += one.triChecks;
                st.visitHash = rays == 1 ? one.visitHash : (st.visitHash ^ one.visitHash) * 16777619u;
            }
        }
        return incomingLight;
    }
    // RI:503-559
    void TraceRays(uint32_t idx, uint32_t idy, orc_counters& cnt) const {
        DebugData debugData = {0, 0};
        uint32_t state = idy * (uint32_t)p.Resolution[0] + idx + (uint32_t)p.FrameRand;
        uint32_t rayIndex = ((uint32_t)p.Resolution[0] * idy + idx) * (uint32_t)p.RaysPerPixel;
        int64_t px = (int64_t)idy * W + idx;
        cnt.activePixels++;
        rt_ris_hit_info hi = ld(f.HitInfos, f.nHitInfos, (int64_t)(idy * (uint32_t)p.Resolution[0] + idx));
        HitInfo hitInfo;
        hitInfo.dst = hi.dst; hitInfo.hitPoint = F3(hi.hitPoint); hitInfo.uv = {hi.uv.x, hi.uv.y};
        hitInfo.normal = F3(hi.normal); hitInfo.materialIndex = hi.materialIndex;
        rt_float4 RISData = RayDirFromCandidates(rayIndex, state);
        f3 risDir = F3(RISData.x, RISData.y, RISData.z);
        if (hitInfo.dst == INF) {
            storeRT(f.Result, px, {0, 0, 0, 0});
            return;
        } else if (anyZero(risDir)) {
            rt_ris_material material = ld(s.Materials, s.nMaterials, hitInfo.materialIndex);
            f3 col = F3(material.color) * material.brightness / sqr(hitInfo.dst);
            storeRT(f.Result, px, {col.x, col.y, col.z, 0});
            return;
        }
        Ray ray;
        ray.pos = hitInfo.hitPoint;
        ray.dir = risDir;
        float correctionWeight = RISData.w * (float)p.RaysPerPixel;
        f3 combined = F3(0, 0, 0);
        RayStats st;
        uint64_t rays = 0;
        for (int i = 0; i < p.RaysPerPixel; i++) combined = combined + Trace(ray, hitInfo, debugData, state, st, rays);
        combined = combined / correctionWeight;
        storeRT(f.Result, px, {combined.x, combined.y, combined.z, 0});
        if (f.records && rays) f.records[px].bounceVisitHash = st.visitHash;
        cnt.rays += rays; cnt.pops += st.pops; cnt.bvChecks += st.bvChecks; cnt.objVisits += st.objVisits;
        cnt.triIters += st.triIters; cnt.triChecks += st.triChecks;
    }
};

template <class Fn>
int forEachPixel(const orc_ris_scene* s, const orc_ris_params* p, orc_ris_frame* f, int threads, Fn fn) {
    if (!s || !p || !f) return -1;
    Ctx ctx(*s, *p, *f);
    int W = ctx.W, H = ctx.H;
    int r0 = 0, r1 = H;
    if (p->row1 > 0) { r0 = p->row0; r1 = p->row1 < H ? p->row1 : H; }
    if (r0 < 0) r0 = 0;
    int ty0 = r0 / 8, ty1 = (r1 + 7) / 8, txn = (W + 7) / 8;
    int64_t ntiles = (int64_t)(ty1 - ty0) * txn;
    if (threads <= 0) threads = omp_get_max_threads();
    orc_counters total;
    std::memset(&total, 0, sizeof total);
#pragma omp parallel num_threads(threads)
    {
        orc_counters c;
        std::memset(&c, 0, sizeof c);
#pragma omp for schedule(dynamic, 8)
        for (int64_t t = 0; t < ntiles; t++) {
            int ty = ty0 + (int)(t / txn), tx = (int)(t % txn);
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++) {
                    int x = tx * 8 + lx, y = ty * 8 + ly;
                    if (x >= W || y >= H || y < r0 || y >= r1) continue;
                    fn(ctx, (uint32_t)x, (uint32_t)y, c);
                }
        }
#pragma omp critical
        {
            total.rays += c.rays; total.pops += c.pops; total.bvChecks += c.bvChecks; total.objVisits += c.objVisits;
            total.triIters += c.triIters; total.triChecks += c.triChecks; total.activePixels += c.activePixels;
        }
    }
    if (f->counters) {
        orc_counters* d = f->counters;
        d->rays += total.rays; d->pops += total.pops; d->bvChecks += total.bvChecks; d->objVisits += total.objVisits;
        d->triIters += total.triIters; d->triChecks += total.triChecks; d->activePixels += total.activePixels;
    }
    return 0;
}

}  // namespace

extern "C" {

// PreCalc.compute:18-28,30-41 (RIS project): the normal is taken to world space with the parent object's matrix.
int orc_ris_calc_tri_normals(orc_ris_scene* s, uint32_t trisNum, int threads) {
    if (!s) return -1;
    if (threads <= 0) threads = omp_get_max_threads();
    int64_t n = (int64_t)trisNum < s->nTris ? (int64_t)trisNum : s->nTris;
#pragma omp parallel for num_threads(threads)
    for (int64_t i = 0; i < n; i++) {
        rt_ris_tri tri = s->Tris[i];
        f3 ab = F3(tri.vB) - F3(tri.vA);
        f3 ac = F3(tri.vC) - F3(tri.vA);
        f3 localNormal = normalize(cross(ab, ac));
        rt_scene_object parent = ld(s->SceneObjects, s->nSceneObjects, tri.parentIndex);
        tri.worldNormal = R3(normalize(mul44(parent.localToWorldMatrix, localNormal, 0)));
        s->Tris[i] = tri;
    }
    return 0;
}
int orc_ris_generate_candidates(const orc_ris_scene* s, const orc_ris_params* p, orc_ris_frame* f, int threads) {
    return forEachPixel(s, p, f, threads, [](const Ctx& c, uint32_t x, uint32_t y, orc_counters& n) { c.GenerateCandidates(x, y, n); });
}
int orc_ris_trace_rays(const orc_ris_scene* s, const orc_ris_params* p, orc_ris_frame* f, int threads) {
    return forEachPixel(s, p, f, threads, [](const Ctx& c, uint32_t x, uint32_t y, orc_counters& n) { c.TraceRays(x, y, n); });
}

int32_t orc_ris_weighted_rand(float a, float b, uint32_t* state) { return weightedRandRIS(a, b, *state) ? 1 : 0; }
float orc_ris_ray_bv_intersect(const float pos[3], const float dir[3], const rt_bv* bv) {
    Ray r = {F3(pos[0], pos[1], pos[2]), F3(dir[0], dir[1], dir[2])};
    return Ctx::RayBVIntersect(r, *bv);
}
int32_t orc_ris_ray_triangle_intersect(const float pos[3], const float dir[3], const rt_ris_tri* tri, float* dst, float uv[2]) {
    Ray r = {F3(pos[0], pos[1], pos[2]), F3(dir[0], dir[1], dir[2])};
    f2 o;
    bool hit = Ctx::RayTriangleIntersect(r, *tri, *dst, o);
    uv[0] = o.x; uv[1] = o.y;
    return hit ? 1 : 0;
}

}  // extern "C"
