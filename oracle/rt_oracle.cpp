// rt_oracle.cpp — CPU restatement of the ReSTIR project's compute kernels.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  PARITY UNPINNED against the reference's own execution: the
// reference ships no golden vectors and its HLSL cannot run outside Unity.  Every function below cites the
// reference lines it restates; paths are relative to BVH_ReSTIR_DL/Assets/Scripts/ComputeShaders/.
//
// Float conventions (there is no HLSL compiler here to ask, so these DEFINE the arithmetic):
//   fp32 throughout, no FMA contraction (build with -ffp-contract=off), IEEE divide and sqrt;
//   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;  mul(M,v) = per row, products summed left to right;
//   normalize(v) = v * (1 / sqrt(dot(v,v)));  lerp(x,y,s) = x + s*(y-x);  min/max ignore a NaN operand;
//   sin/cos/log/pow/acos/asin/atan2 are the polynomial routines of include/rt_detmath.h;
//   out-of-range buffer and texture reads return 0 and out-of-range writes are dropped (D3D robust access).
#include "rt_oracle.h"
#include "orc_math.h"

#include <omp.h>
#include <cstring>
#include <vector>

namespace {

using namespace orcm;

// ---------------- BRDF.hlsl ----------------
inline float FresnelSchlickX(float cosTheta, float F0x) {  // :1-4, x channel
    return F0x + (1.0f - F0x) * rtdm::pow_(1.0f - cosTheta, 5.0f);
}
inline float GeometrySmith(float NdotL, float NdotV, float roughness) {  // :6-13
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    float G1 = NdotL / (NdotL * (1.0f - k) + k);
    float G2 = NdotV / (NdotV * (1.0f - k) + k);
    return G1 * G2;
}
inline float DistributionGGX(float NdotH, float roughness) {  // :15-23
    float a = roughness * roughness;
    float a2 = a * a;
    float NdotH2 = NdotH * NdotH;
    float denom = (NdotH2 * (a2 - 1.0f) + 1.0f);
    return a2 / (3.14f * denom * denom);
}
// :25-65.  The function is declared `float` but returns a float3 expression, so HLSL keeps the x channel;
// with hitAlbedo = (1,0,1) and metallicity forced to 1 only the x lane is evaluated here.
float GetBRDF(f3 incomingDir, f3 outgoingDir, f3 hitNormal, float metallicity, float roughness) {
    metallicity = 1;
    incomingDir = -incomingDir;
    const float hitAlbedoX = 1.0f;
    f3 H = normalize(incomingDir + outgoingDir);
    float NdotL = saturate(dot(hitNormal, incomingDir));
    float NdotV = saturate(dot(hitNormal, outgoingDir));
    float NdotH = saturate(dot(hitNormal, H));
    float VdotH = saturate(dot(outgoingDir, H));
    float F0x = 0.04f + metallicity * (hitAlbedoX - 0.04f);  // lerp(0.04, albedo, metallicity)
    float Fx = FresnelSchlickX(VdotH, F0x);
    float D = DistributionGGX(NdotH, roughness);
    float G = GeometrySmith(NdotL, NdotV, roughness);
    float numeratorX = D * G * Fx;
    float denominator = 4.0f * NdotL * NdotV + 0.001f;
    float specularX = numeratorX / denominator;
    float kSx = Fx;
    float kDx = 1.0f - kSx;
    kDx *= 1.0f - metallicity;
    float diffuseX = kDx * hitAlbedoX / 3.14f;
    return (diffuseX + specularX) * NdotL * 10;
}

// ---------------- ray-tracer structs: RendererDataTypes.hlsl:73-123 ----------------
struct Ray { f3 pos, dir, invDir; };
struct HitInfo { float dst; f3 hitPoint; f2 uv; f3 normal; f3 incomingDir; int materialIndex; };
struct DebugData { int triChecks, bvChecks; };
struct RayStats {  // not in the reference: what the trace records and counters are built from
    uint32_t visitHash = 2166136261u;
    int triIndex = -1, objIndex = -1;
    uint64_t pops = 0, bvChecks = 0, objVisits = 0, triIters = 0, triChecks = 0;
    int curTri = -1;
    inline void visitInternal(uint32_t node) { visitHash = (visitHash ^ (0x80000000u | node)) * 16777619u; }
    inline void visitLeaf(int a, int b) {
        visitHash = (visitHash ^ (uint32_t)a) * 16777619u;
        visitHash = (visitHash ^ (uint32_t)b) * 16777619u;
    }
};

inline HitInfo InitHitInfo() {  // RendererDataTypes.hlsl:154-165
    HitInfo h;
    h.dst = INF;
    h.hitPoint = F3(-1, -1, -1);
    h.uv = {0, 0};
    h.normal = F3(0, 0, 0);
    h.incomingDir = F3(0, 0, 0);
    h.materialIndex = 0;
    return h;
}
inline rt_reservoir EmptyReservoir() {  // :218-230
    rt_reservoir r;
    std::memset(&r, 0, sizeof r);
    return r;
}
inline rt_material InitMaterial() {  // :174-204
    rt_material m;
    std::memset(&m, 0, sizeof m);
    return m;
}

struct Ctx {
    const orc_scene& s;
    const orc_params& p;
    orc_frame& f;
    int W, H;
    float atlasDims[2], envDims[2];
    Ctx(const orc_scene& s_, const orc_params& p_, orc_frame& f_) : s(s_), p(p_), f(f_) {
        W = p.Resolution[0];
        H = p.Resolution[1];
        for (int i = 0; i < 2; i++) {
            atlasDims[i] = rtdm::bits2f(p.TextureAtlasDimsRaw[i]);
            envDims[i] = rtdm::bits2f(p.EnvironmentMapTexDimsRaw[i]);
        }
    }
    int64_t npx() const { return (int64_t)W * H; }

    // Texture2D.Load with robust out-of-range behaviour.
    static f3 texLoad(const rt_float4* tex, int w, int h, uint32_t x, uint32_t y) {
        if (!tex || x >= (uint32_t)w || y >= (uint32_t)h) return F3(0, 0, 0);
        rt_float4 c = tex[(int64_t)y * w + x];
        return F3(c.x, c.y, c.z);
    }
    // ReSTIR.compute:94-100
    f3 SampleTexture(f2 uv, uint32_t locX, uint32_t locY, uint32_t dimX, uint32_t dimY, const rt_float4* tex, int w,
                     int h) const {
        uint32_t tx = locX + f2u(uv.x * (float)dimX);
        uint32_t ty = locY + f2u(uv.y * (float)dimY);
        return texLoad(tex, w, h, tx, ty);
    }
    f3 SampleAtlas(f2 uv, rt_int2 loc, rt_int2 dims) const {
        return SampleTexture(uv, (uint32_t)loc.x, (uint32_t)loc.y, (uint32_t)dims.x, (uint32_t)dims.y, s.TextureAtlas,
                             s.atlasW, s.atlasH);
    }
    // :102-117.  EnvironmentMapTexDims is a float2 constant converted to the uint2 parameter.
    f3 SampleEnvironmentMap(f3 dir) const {
        dir = -dir;
        float theta = rtdm::atan2_(dir.z, dir.x);
        float phi = rtdm::asin_(dir.y);
        float u = (theta / (2.0f * PI)) + 0.5f;
        float v = 0.5f - (phi / PI);
        return SampleTexture({u, v}, 0, 0, f2u(envDims[0]), f2u(envDims[1]), s.EnvironmentMap, s.envW, s.envH);
    }

    // :119-146
    bool RayTriangleIntersect(const Ray& ray, const rt_triangle& tri, float& dstOut, f2& uvOut) const {
        rt_vertex v0 = ld(s.Vertices, s.nVertices, tri.vertex0Index);
        rt_vertex v1 = ld(s.Vertices, s.nVertices, tri.vertex1Index);
        rt_vertex v2 = ld(s.Vertices, s.nVertices, tri.vertex2Index);
        f3 edgeAB = F3(v1.pos) - F3(v0.pos);
        f3 edgeAC = F3(v2.pos) - F3(v0.pos);
        f3 normalVector = cross(edgeAB, edgeAC);
        f3 ao = ray.pos - F3(v0.pos);
        f3 dao = cross(ao, ray.dir);
        float determinant = -dot(ray.dir, normalVector);
        float invDet = 1 / determinant;
        float dst = dot(ao, normalVector) * invDet;
        float u = dot(edgeAC, dao) * invDet;
        float v = -dot(edgeAB, dao) * invDet;
        float w = 1 - u - v;
        bool didHit = determinant >= 1E-8f && dst >= 0 && u >= 0 && v >= 0 && w >= 0;
        if (didHit) {
            uvOut.x = (v0.uv.x * w + v1.uv.x * u) + v2.uv.x * v;
            uvOut.y = (v0.uv.y * w + v1.uv.y * u) + v2.uv.y * v;
        } else {
            uvOut = {0, 0};
        }
        dstOut = dst;
        return didHit;
    }
    // :148-161
    static float RayBVIntersect(const Ray& ray, const rt_bv& bv) {
        f3 tMin = (F3(bv.min) - ray.pos) * ray.invDir;
        f3 tMax = (F3(bv.max) - ray.pos) * ray.invDir;
        f3 t1 = min3(tMin, tMax);
        f3 t2 = max3(tMin, tMax);
        float tNear = fmaxf_(fmaxf_(t1.x, t1.y), t1.z);
        float tFar = fminf_(fminf_(t2.x, t2.y), t2.z);
        bool didHit = tFar >= tNear && tFar > 0;
        return didHit ? (tNear > 0 ? tNear : 0) : INF;
    }
    // :163-217
    void TraverseTrisBVH(HitInfo& hitInfo, DebugData& debugData, const Ray& ray, uint32_t bvStartIndex,
                         RayStats& st) const {
        uint32_t bvStack[MAX_BVH_DEPTH + 1 + 64];  // the reference overflows at 33; the slack keeps the oracle defined
        uint32_t stackPointer = 0;
        bvStack[stackPointer++] = bvStartIndex;
        while (stackPointer > 0) {
            uint32_t node = bvStack[--stackPointer];
            rt_bv bv = ld(s.BVs, s.nBVs, node);
            st.pops++;
            bool isLeaf = bv.indexA >= 0;
            if (isLeaf) st.visitLeaf(bv.indexA, bv.indexB); else st.visitInternal(node);
            bv.indexA = bv.indexA < 0 ? -bv.indexA : bv.indexA;
            bv.indexB = bv.indexB < 0 ? -bv.indexB : bv.indexB;
            if (isLeaf) {
                for (int triIndex = bv.indexA; triIndex < bv.indexA + bv.indexB; triIndex++) {
                    rt_triangle tri = ld(s.Triangles, s.nTriangles, triIndex);
                    st.triIters++;
                    if (dot(F3(tri.localNormal), ray.dir) <= 0) {
                        float dst; f2 uv;
                        bool didHit = RayTriangleIntersect(ray, tri, dst, uv);
                        debugData.triChecks++;
                        st.triChecks++;
                        if (didHit && dst < hitInfo.dst) {
                            hitInfo.dst = dst;
                            hitInfo.uv = uv;
                            hitInfo.normal = F3(tri.localNormal);
                            st.curTri = triIndex;
                        }
                    }
                }
            } else {
                rt_bv bvA = ld(s.BVs, s.nBVs, bv.indexA);
                rt_bv bvB = ld(s.BVs, s.nBVs, bv.indexB);
                float dstA = RayBVIntersect(ray, bvA);
                float dstB = RayBVIntersect(ray, bvB);
                debugData.bvChecks += 2;
                st.bvChecks += 2;
                if (stackPointer + 2 > MAX_BVH_DEPTH + 1 + 64) continue;
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
    // :219-282
    void TraverseSceneBVH(HitInfo& hitInfo, DebugData& debugData, const Ray& ray, RayStats& st) const {
        uint32_t bvStack[MAX_BVH_DEPTH + 1 + 64];
        uint32_t stackPointer = 0;
        bvStack[stackPointer++] = p.SceneBVHStartIndex;
        while (stackPointer > 0) {
            uint32_t node = bvStack[--stackPointer];
            rt_bv bv = ld(s.BVs, s.nBVs, node);
            st.pops++;
            bool isLeaf = bv.indexA >= 0;
            if (isLeaf) st.visitLeaf(bv.indexA, bv.indexB); else st.visitInternal(node);
            bv.indexA = bv.indexA < 0 ? -bv.indexA : bv.indexA;
            bv.indexB = bv.indexB < 0 ? -bv.indexB : bv.indexB;
            if (isLeaf) {
                for (int sceneObjectIndex = bv.indexA; sceneObjectIndex < bv.indexA + bv.indexB; sceneObjectIndex++) {
                    rt_scene_object sceneObject = ld(s.SceneObjects, s.nSceneObjects, sceneObjectIndex);
                    st.objVisits++;
                    Ray localRay;
                    localRay.pos = mul44(sceneObject.worldToLocalMatrix, ray.pos, 1);
                    localRay.dir = mul44(sceneObject.worldToLocalMatrix, ray.dir, 0);
                    localRay.invDir = F3(1 / localRay.dir.x, 1 / localRay.dir.y, 1 / localRay.dir.z);
                    HitInfo localHitInfo = InitHitInfo();
                    st.curTri = -1;
                    TraverseTrisBVH(localHitInfo, debugData, localRay, (uint32_t)sceneObject.bvStartIndex, st);
                    rt_material material = ld(s.Materials, s.nMaterials, sceneObject.materialIndex);
                    if (material.normalsTexLoc.x != -1 && material.normalsTexLoc.y != -1)
                        localHitInfo.normal =
                            SampleAtlas(localHitInfo.uv, material.normalsTexLoc, material.normalsTexDims) * 2 - 1;
                    if (localHitInfo.dst < hitInfo.dst) {
                        hitInfo.dst = localHitInfo.dst;
                        hitInfo.normal = normalize(mul44(sceneObject.localToWorldMatrix, localHitInfo.normal, 0));
                        hitInfo.hitPoint = ray.pos + ray.dir * localHitInfo.dst;
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
                if (stackPointer + 2 > MAX_BVH_DEPTH + 1 + 64) continue;
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
    // :284-291
    HitInfo GetRayCollision(const Ray& ray, DebugData& debugData, RayStats& st) const {
        HitInfo hitInfo = InitHitInfo();
        TraverseSceneBVH(hitInfo, debugData, ray, st);
        return hitInfo;
    }

    // :293-313
    int LightObjectReservoirSample(uint32_t& state) const {
        int firstIndex = randIntSpan(0, (int)p.EmittingObjectsNum, state);
        float firstWeight = ld(s.LightObjects, s.nLightObjects, firstIndex).areaApprox;
        int chosenIndex = firstIndex;
        float totWeights = firstWeight;
        for (uint32_t i = 0; i < p.SceneObjectReservoirTestsNum; i++) {
            int candidateIndex = randIntSpan(0, (int)p.EmittingObjectsNum, state);
            float candidateWeight = ld(s.LightObjects, s.nLightObjects, candidateIndex).areaApprox;
            if (weightedRand(candidateWeight, totWeights, state)) chosenIndex = candidateIndex;
            totWeights += candidateWeight;
        }
        return chosenIndex;
    }
    // :328-348
    int TriReservoirSample(uint32_t startTriIndex, uint32_t endTriIndex, uint32_t& state) const {
        int firstIndex = randIntSpan((int)startTriIndex, (int)endTriIndex, state);
        float firstWeight = ld(s.Triangles, s.nTriangles, firstIndex).area;
        int chosenIndex = firstIndex;
        float totWeights = firstWeight;
        for (uint32_t i = 0; i < p.TriReservoirTestsNum; i++) {
            int candidateIndex = randIntSpan((int)startTriIndex, (int)endTriIndex, state);
            float candidateWeight = ld(s.Triangles, s.nTriangles, candidateIndex).area;
            if (weightedRand(candidateWeight, totWeights, state)) chosenIndex = candidateIndex;
            totWeights += candidateWeight;
        }
        return chosenIndex;
    }
    // :351-360
    f2 GenerateCandidateWeight(float sqrDst, float hitAngleDot, f3 incomingDir, f3 outgoingDir, f3 hitNormal,
                               float metallicity, float roughness, float brightness) const {
        float weight = brightness * hitAngleDot / sqrDst;
        float brdf = 1.0f;
        if (p.kw_BRDF) {
            brdf = GetBRDF(incomingDir, outgoingDir, hitNormal, metallicity, roughness);
            weight *= brdf;
        }
        return {weight, brdf};
    }
    // :362-383
    f3 GetRandWorldPointTri(const rt_triangle& tri, const rt_float4x4& localToWorldMatrix, uint32_t& state) const {
        f3 v0 = F3(ld(s.Vertices, s.nVertices, tri.vertex0Index).pos);
        f3 v1 = F3(ld(s.Vertices, s.nVertices, tri.vertex1Index).pos);
        f3 v2 = F3(ld(s.Vertices, s.nVertices, tri.vertex2Index).pos);
        float r1 = randNormalized(state);
        float r2 = randNormalized(state);
        if (r1 + r2 > 1.0f) {
            r1 = 1.0f - r1;
            r2 = 1.0f - r2;
        }
        f3 localPoint = v0 * (1.0f - r1 - r2) + v1 * r1 + v2 * r2;
        return mul44(localToWorldMatrix, localPoint, 1.0f);
    }
    float matMetallicity(const rt_material& m, f2 uv) const {
        return m.metallicity == -1 ? SampleAtlas(uv, m.metallicityTexLoc, m.metallicityTexDims).x : m.metallicity;
    }
    float matRoughness(const rt_material& m, f2 uv) const {
        return m.roughness == -1 ? SampleAtlas(uv, m.roughnessTexLoc, m.roughnessTexDims).x : m.roughness;
    }
    f3 matCol(const rt_material& m, f2 uv) const {
        bool tex = m.col.x == -1 && m.col.y == -1 && m.col.z == -1;
        return tex ? SampleAtlas(uv, m.colTexLoc, m.colTexDims) : F3(m.col);
    }

    // :385-434
    void GenerateCandidatesDL(const HitInfo& hitInfo, uint32_t pixelIndex, uint32_t& state, orc_trace_record* rec,
                              RayStats& shadowStats, bool& castShadow) const {
        uint32_t lightObjectIndex = (uint32_t)LightObjectReservoirSample(state);
        rt_light_object lightObject = ld(s.LightObjects, s.nLightObjects, lightObjectIndex);
        uint32_t startTriIndex = (uint32_t)lightObject.triStart;
        uint32_t endTriIndex = (uint32_t)(lightObject.triStart + lightObject.totTris);
        uint32_t triIndex = (uint32_t)TriReservoirSample(startTriIndex, endTriIndex, state);
        if (rec) { rec->lightObj = (int)lightObjectIndex; rec->lightTri = (int)triIndex; }
        rt_triangle tri = ld(s.Triangles, s.nTriangles, triIndex);
        f3 worldPoint = GetRandWorldPointTri(tri, lightObject.localToWorldMatrix, state);
        f3 dst = worldPoint - hitInfo.hitPoint;
        f3 dir = normalize(dst);
        if (dot(hitInfo.normal, dir) < 0) {
            if (pixelIndex < (uint64_t)npx()) f.Candidates[pixelIndex] = EmptyReservoir();
            return;
        }
        f3 worldNormal = normalize(mul44(lightObject.localToWorldMatrix, F3(tri.localNormal), 0));
        float hitAngleDot = fmaxf_(dot(worldNormal, dir), 0);
        float sqrDst = dot2(dst);
        rt_material material = ld(s.Materials, s.nMaterials, hitInfo.materialIndex);
        float metallicity = matMetallicity(material, hitInfo.uv);
        float roughness = matRoughness(material, hitInfo.uv);
        f2 wb = GenerateCandidateWeight(sqrDst, hitAngleDot, hitInfo.incomingDir, dir, hitInfo.normal, metallicity,
                                        roughness, lightObject.brightness);
        if (p.kw_VIS_REUSE) {
            Ray shadowRay;  // InitRay(pos, dir), RendererDataTypes.hlsl:136-144
            shadowRay.pos = hitInfo.hitPoint;
            shadowRay.dir = dir;
            shadowRay.invDir = F3(1 / dir.x, 1 / dir.y, 1 / dir.z);
            DebugData t = {0, 0};
            castShadow = true;
            HitInfo shadowHitInfo = GetRayCollision(shadowRay, t, shadowStats);
            if (ld(s.Materials, s.nMaterials, shadowHitInfo.materialIndex).brightness == 0) wb.x = 0;
        }
        rt_reservoir r;
        r.dir = R3(dir);
        r.hitPoint = R3(hitInfo.hitPoint);
        r.normal = R3(hitInfo.normal);
        r.brdf = wb.y;
        r.chosenWeight = wb.x;
        r.totWeights = wb.x;
        r.totCandidates = 1;
        if (pixelIndex < (uint64_t)npx()) f.Candidates[pixelIndex] = r;
    }

    uint32_t GetPixelIndex(uint32_t x, uint32_t y) const { return (uint32_t)p.Resolution[0] * y + x; }  // :436-439
    // :441-449
    f3 GetFocalPos(f2 uv) const {
        float focalPosX = (uv.x - 0.5f) * 2.0f * p.ViewSpaceDims[0];
        float focalPosY = (uv.y - 0.5f) * 2.0f * p.ViewSpaceDims[1];
        f3 focalPos = F3(focalPosX, focalPosY, 1) * p.FocalPlaneFactor;
        f3 cam = F3(p.WorldSpaceCameraPos[0], p.WorldSpaceCameraPos[1], p.WorldSpaceCameraPos[2]);
        return mul33(p.CameraTransform, focalPos) + cam;
    }
    // :451-461
    f2 GetLastUV(f3 worldFocalPos) const {
        f3 last = F3(p.LastWorldSpaceCameraPos[0], p.LastWorldSpaceCameraPos[1], p.LastWorldSpaceCameraPos[2]);
        f3 focalPos = worldFocalPos - last;
        f3 cameraFocalPos = mul33(p.LastCameraTransformInverse, focalPos);
        f3 ndcFocalPos = cameraFocalPos / p.FocalPlaneFactor;
        float uvX = (ndcFocalPos.x / (2.0f * p.ViewSpaceDims[0])) + 0.5f;
        float uvY = (ndcFocalPos.y / (2.0f * p.ViewSpaceDims[1])) + 0.5f;
        return {clampf(uvX, 0.0f, 1.0f), clampf(uvY, 0.0f, 1.0f)};
    }
    // Store to a 3- or 4-channel render target.  With 8-bit targets (TextureHelper.cs:144-155: RenderTextureFormat.Default)
    // the value is converted as D3D/Metal convert float to UNORM8: NaN -> 0, clamp to [0, 1], scale by 255, add 0.5,
    // truncate; it is kept as the float the texture unit would return on the next load (n / 255).
    static float toUnorm8(float x) { return (float)(int)(saturate(x) * 255.0f + 0.5f) / 255.0f; }
    void storeRT(rt_float4* tex, int64_t px, rt_float4 v) const {
        if (p.unorm8Targets) v = {toUnorm8(v.x), toUnorm8(v.y), toUnorm8(v.z), toUnorm8(v.w)};
        tex[px] = v;
    }
    float depthAt(int x, int y) const {  // RWTexture2D read, out of range -> 0
        if (x < 0 || y < 0 || x >= W || y >= H) return 0.0f;
        return f.DepthBuffer[(int64_t)y * W + x];
    }

    // ---- kernel InitialTrace, :463-510 ----
    void InitialTrace(uint32_t idx, uint32_t idy, orc_counters& cnt) const {
        DebugData debugData = {0, 0};
        uint32_t state = idy * (uint32_t)p.Resolution[0] + idx + p.FrameRand;
        f2 uv = {(float)idx / (float)p.Resolution[0], (float)idy / (float)p.Resolution[1]};
        f3 worldFocalPos = GetFocalPos(uv);
        uint32_t pixelIndex = GetPixelIndex(idx, idy);
        Ray ray;
        f2 c = randPointInCircle(state) * p.DefocusStrength;
        f3 jitter = mul33(p.CameraTransform, F3(c.x, c.y, 0));
        f3 cam = F3(p.WorldSpaceCameraPos[0], p.WorldSpaceCameraPos[1], p.WorldSpaceCameraPos[2]);
        ray.pos = cam + jitter * p.FocalPlaneFactor;
        ray.dir = normalize(worldFocalPos - ray.pos);
        ray.invDir = F3(1 / ray.dir.x, 1 / ray.dir.y, 1 / ray.dir.z);

        RayStats st, shadowSt;
        HitInfo hitInfo = GetRayCollision(ray, debugData, st);
        bool didHit = hitInfo.dst != INF;
        hitInfo.incomingDir = ray.dir;

        rt_material material = ld(s.Materials, s.nMaterials, hitInfo.materialIndex);
        float bump = material.bump == -1 ? SampleAtlas(hitInfo.uv, material.bumpTexLoc, material.bumpTexDims).x
                                         : material.bump;
        hitInfo.hitPoint = hitInfo.hitPoint - hitInfo.incomingDir * bump;

        orc_trace_record* rec = f.records ? &f.records[pixelIndex] : nullptr;
        if (rec) {
            rec->triIndex = st.triIndex; rec->objIndex = st.objIndex; rec->visitHash = st.visitHash;
            rec->lightObj = -1; rec->lightTri = -1; rec->shadowVisitHash = 0; rec->shadowTri = -1;
            rec->bounceVisitHash = 0;
        }
        bool castShadow = false;
        if (didHit) GenerateCandidatesDL(hitInfo, pixelIndex, state, rec, shadowSt, castShadow);
        if (rec && castShadow) { rec->shadowVisitHash = shadowSt.visitHash; rec->shadowTri = shadowSt.triIndex; }

        rt_hit_info out;
        out.dst = hitInfo.dst; out.hitPoint = R3(hitInfo.hitPoint); out.uv = {hitInfo.uv.x, hitInfo.uv.y};
        out.normal = R3(hitInfo.normal); out.incomingDir = R3(hitInfo.incomingDir);
        out.materialIndex = hitInfo.materialIndex;
        f.HitInfos[pixelIndex] = out;

        bool badMaxBVHDepth = p.MaxBVHDepth > MAX_BVH_DEPTH;
        int64_t px = (int64_t)idy * W + idx;
        storeRT(f.DebugOverlay, px, {debugData.triChecks / (float)p.DebugDataMaxValues[0],
                                     debugData.bvChecks / (float)p.DebugDataMaxValues[1], badMaxBVHDepth ? 1.0f : 0.0f, 0});
        f.DepthBuffer[px] = hitInfo.dst;
        f3 n = (hitInfo.normal + 1) * 0.5f;
        storeRT(f.NormalsBuffer, px, {n.x, n.y, n.z, 0});
        f3 firstHit = didHit ? hitInfo.hitPoint : F3(0, 0, 0);
        storeRT(p.FrameCount % 2 == 0 ? f.RayHitPointsA : f.RayHitPointsB, px, {firstHit.x, firstHit.y, firstHit.z, 0});

        cnt.rays += 1 + (castShadow ? 1 : 0);
        cnt.pops += st.pops + shadowSt.pops; cnt.bvChecks += st.bvChecks + shadowSt.bvChecks;
        cnt.objVisits += st.objVisits + shadowSt.objVisits; cnt.triIters += st.triIters + shadowSt.triIters;
        cnt.triChecks += st.triChecks + shadowSt.triChecks; cnt.activePixels++;
    }

    const rt_reservoir* srcBuf() const { return p.ReuseBufferCycle ? f.Candidates : f.CandidatesB; }
    rt_reservoir* dstBuf() const { return p.ReuseBufferCycle ? f.CandidatesB : f.Candidates; }

    // :514-556
    rt_reservoir SpatialCandidateReservoirSample(uint32_t pixelBaseIndex, int tx, int ty, uint32_t& state,
                                                 orc_counters& cnt) const {
        rt_reservoir baseReservoir = ld(srcBuf(), npx(), pixelBaseIndex);
        rt_reservoir spatialReservoir = baseReservoir;
        for (uint32_t i = 0; i < p.TemporalCandidatesNum; i++) {
            int ox = f2i((float)randIntSpan(-p.MaxOffset, p.MaxOffset, state));
            int oy = f2i((float)randIntSpan(-p.MaxOffset, p.MaxOffset, state));
            int offX = tx + ox, offY = ty + oy;
            // `any(offsetID) < 0` compares a bool with 0 and is always false: negative coordinates pass.
            if (offX >= p.Resolution[0] || offY >= p.Resolution[1] || depthAt(offX, offY) == INF) continue;
            uint32_t pixelIndex = GetPixelIndex((uint32_t)offX, (uint32_t)offY);
            rt_reservoir cand = ld(srcBuf(), npx(), pixelIndex);
            cnt.validNeighbours++;
            // `all(dir) == 0`: true as soon as ANY component is exactly zero.
            if (cand.dir.x == 0 || cand.dir.y == 0 || cand.dir.z == 0) continue;
            float hitPointDiff = length(F3(baseReservoir.hitPoint) - F3(cand.hitPoint));
            float normalDiffDeg = angleBetweenNormals(F3(baseReservoir.normal), F3(cand.normal));
            if (hitPointDiff > p.SpatialHitPointDiffThreshold || normalDiffDeg > p.SpatialNormalsAngleThreshold ||
                rtdm::fabs_(spatialReservoir.brdf - cand.brdf) > p.SpatialBRDFThreshold)
                continue;
            if (weightedRand(cand.chosenWeight, spatialReservoir.totWeights, state)) {
                spatialReservoir.dir = cand.dir;
                spatialReservoir.chosenWeight = cand.chosenWeight;
                spatialReservoir.hitPoint = cand.hitPoint;
                spatialReservoir.normal = cand.normal;
                spatialReservoir.brdf = cand.brdf;
            }
            spatialReservoir.totWeights += cand.totWeights;
            spatialReservoir.totCandidates += cand.totCandidates;
        }
        return spatialReservoir;
    }
    // ---- kernel SpatialReusePass, :558-571 ----
    void SpatialReusePass(uint32_t idx, uint32_t idy, orc_counters& cnt) const {
        if (depthAt((int)idx, (int)idy) == INF) return;
        uint32_t state = idy * (uint32_t)p.Resolution[0] + idx + p.FrameRand;
        uint32_t pixelIndex = GetPixelIndex(idx, idy);
        rt_reservoir r = SpatialCandidateReservoirSample(pixelIndex, (int)idx, (int)idy, state, cnt);
        dstBuf()[pixelIndex] = r;
        cnt.activePixels++;
    }
    // ---- kernel TransferToOriginal, :575-582 ----
    void TransferToOriginal(uint32_t idx, uint32_t idy, orc_counters& cnt) const {
        if (depthAt((int)idx, (int)idy) == INF) return;
        uint32_t pixelIndex = GetPixelIndex(idx, idy);
        f.Candidates[pixelIndex] = f.CandidatesB[pixelIndex];
        cnt.activePixels++;
    }
    // ---- kernel TemporalReuse, :586-641 ----
    void TemporalReuse(uint32_t idx, uint32_t idy, orc_counters& cnt) const {
        if (depthAt((int)idx, (int)idy) == INF || p.FrameCount == 0) return;
        cnt.activePixels++;
        uint32_t state = idy * (uint32_t)p.Resolution[0] + idx + p.FrameRand;
        uint32_t pixelIndex = GetPixelIndex(idx, idy);
        f2 uv = {(float)idx / (float)p.Resolution[0], (float)idy / (float)p.Resolution[1]};
        f3 worldFocalPos = GetFocalPos(uv);
        f2 lastFrameUV = GetLastUV(worldFocalPos);
        if (length(uv - lastFrameUV) > p.TemporalPrecisionThreshold) {
            f.TemporalFrameBuffer[pixelIndex] = f.Candidates[pixelIndex];
            cnt.temporalCopies++;
            return;
        }
        // else offsetID = id.xy  (:605): the reprojected pixel is discarded, history is always same-pixel.
        uint32_t lastFramePixelIndex = GetPixelIndex(idx, idy);
        int64_t px = (int64_t)idy * W + idx;
        rt_float4 a = f.RayHitPointsA[px], b = f.RayHitPointsB[px];
        float pixelHitPointMovement = length(F3(a.x, a.y, a.z) - F3(b.x, b.y, b.z));
        rt_reservoir thisFrameReservoir = f.Candidates[pixelIndex];
        if (pixelHitPointMovement <= p.PixelMovementThreshold) cnt.temporalMerges++; else cnt.temporalRejects++;
        if (pixelHitPointMovement <= p.PixelMovementThreshold) {
            rt_reservoir lastFrameCandidate = ld(f.TemporalFrameBuffer, npx(), lastFramePixelIndex);
            float adjustedThisFrameWeight = thisFrameReservoir.totWeights * (1.0f - p.TemporalReuseWeight);
            float adjustedLastFrameWeight = lastFrameCandidate.totWeights * p.TemporalReuseWeight;
            float totWeights = adjustedThisFrameWeight + adjustedLastFrameWeight;
            if (weightedRand(adjustedLastFrameWeight, totWeights, state)) {
                thisFrameReservoir.dir = lastFrameCandidate.dir;
                thisFrameReservoir.chosenWeight = lastFrameCandidate.chosenWeight;
                thisFrameReservoir.hitPoint = lastFrameCandidate.hitPoint;
            }
            thisFrameReservoir.totWeights = avg(thisFrameReservoir.totWeights * (1.0f - p.TemporalReuseWeight),
                                                lastFrameCandidate.totWeights * p.TemporalReuseWeight);
            thisFrameReservoir.totCandidates = avg(thisFrameReservoir.totCandidates * (1.0f - p.TemporalReuseWeight),
                                                   lastFrameCandidate.totCandidates * p.TemporalReuseWeight);
        }
        f.Candidates[pixelIndex] = thisFrameReservoir;
        f.TemporalFrameBuffer[pixelIndex] = thisFrameReservoir;
    }

    // ---- kernel TraceRays, :645-815 ----
    struct TraceInfo { float totTravelDst; f3 rayColor, directLight, indirectLight; };
    // `st` sums the counters of all rays cast; its visitHash chains the per-ray hashes.
    TraceInfo Trace(Ray ray, HitInfo hitInfo, uint32_t pixelIndex, DebugData& debugData, uint32_t& state, RayStats& st,
                    uint64_t& rays) const {
        TraceInfo traceInfo;
        traceInfo.totTravelDst = 0;
        float rc = dot(hitInfo.normal, ray.dir);
        traceInfo.rayColor = F3(rc, rc, rc);
        traceInfo.directLight = F3(0, 0, 0);
        traceInfo.indirectLight = F3(0, 0, 0);
        f3 incomingDir = hitInfo.incomingDir;
        for (uint32_t bounceCount = 0; bounceCount <= p.MaxBounceCount; bounceCount++) {
            if (hitInfo.dst == INF) break;
            rt_material material = InitMaterial();  // GetMaterial, :659-664
            if (hitInfo.dst != INF) material = ld(s.Materials, s.nMaterials, hitInfo.materialIndex);
            // IsInvalidCandidate, :666-671
            bool invalid = (bounceCount == 1 && material.brightness <= 0) ||
                           (bounceCount == 0 && dot(hitInfo.normal, ray.dir) < 0) || (hitInfo.dst == INF);
            if (invalid) f.Candidates[pixelIndex] = EmptyReservoir();
            float metallicity = matMetallicity(material, hitInfo.uv);
            float roughness = matRoughness(material, hitInfo.uv);
            if (bounceCount != 0) {
                incomingDir = ray.dir;
                bool isSpecularBounce = p.ScatterProbability < randNormalized(state);
                (void)isSpecularBounce;
                // GetRayDirection / GetScatterDirection, :673-687.  ray.pos and ray.invDir keep their old values.
                f3 newDir = normalize(hitInfo.normal + randPointOnUnitSphere(state));
                float hitAngleDot = dot(hitInfo.normal, newDir);
                if (hitAngleDot < 0) newDir = -newDir;
                ray.dir = newDir;
                if (p.kw_WEIGHT_RECALC && bounceCount == 1) {
                    // UpdateCandidateWeight, :645-657
                    if (material.brightness > 0) {
                        f2 wb = GenerateCandidateWeight(sqr(hitInfo.dst), fmaxf_(dot(hitInfo.normal, ray.dir), 0),
                                                        incomingDir, ray.dir, hitInfo.normal, metallicity, roughness,
                                                        material.brightness);
                        f.TemporalFrameBuffer[pixelIndex].chosenWeight = wb.x;
                        f.TemporalFrameBuffer[pixelIndex].brdf = wb.y;
                    }
                }
            }
            // GetEmittedLight, :689-695
            f3 emittedLight = matCol(material, hitInfo.uv) * material.brightness / sqr(hitInfo.dst);
            // GetRayColFactor, :697-708
            f3 rayColFactor = matCol(material, hitInfo.uv);
            if (p.kw_BRDF)
                rayColFactor = rayColFactor * GetBRDF(incomingDir, ray.dir, hitInfo.normal, metallicity, roughness);
            traceInfo.rayColor = traceInfo.rayColor * rayColFactor;
            traceInfo.totTravelDst += hitInfo.dst;
            if (bounceCount == 0) traceInfo.directLight = traceInfo.directLight + emittedLight * traceInfo.rayColor;
            else traceInfo.indirectLight = traceInfo.indirectLight + emittedLight * traceInfo.rayColor;
            if (bounceCount != p.MaxBounceCount) {
                RayStats one;
                hitInfo = GetRayCollision(ray, debugData, one);
                rays++;
                st.pops += one.pops; st.bvChecks += one.bvChecks; st.objVisits += one.objVisits;
                st.triIters += one.triIters; st.triChecks += one.triChecks;
                st.visitHash = rays == 1 ? one.visitHash : (st.visitHash ^ one.visitHash) * 16777619u;
            }
        }
        return traceInfo;
    }
    void TraceRays(uint32_t idx, uint32_t idy, orc_counters& cnt) const {
        DebugData debugData = {0, 0};
        uint32_t state = idy * (uint32_t)p.Resolution[0] + idx + p.FrameRand;
        uint32_t pixelIndex = GetPixelIndex(idx, idy);
        int64_t px = (int64_t)idy * W + idx;
        cnt.activePixels++;
        rt_hit_info hi = f.HitInfos[idy * (uint32_t)p.Resolution[0] + idx];
        HitInfo hitInfo;
        hitInfo.dst = hi.dst; hitInfo.hitPoint = F3(hi.hitPoint); hitInfo.uv = {hi.uv.x, hi.uv.y};
        hitInfo.normal = F3(hi.normal); hitInfo.incomingDir = F3(hi.incomingDir); hitInfo.materialIndex = hi.materialIndex;
        rt_reservoir candidateReservoir = f.Candidates[pixelIndex];
        f3 cdir = F3(candidateReservoir.dir);
        if (hitInfo.dst == INF) {
            f3 col = SampleEnvironmentMap(hitInfo.incomingDir);
            storeRT(f.Result, px, {col.x, col.y, col.z, 0});
            return;
        } else if ((cdir.x == 0 || cdir.y == 0 || cdir.z == 0) || dot(hitInfo.normal, cdir) < 0) {
            rt_material material = ld(s.Materials, s.nMaterials, hitInfo.materialIndex);
            f3 col = matCol(material, hitInfo.uv) * material.brightness / sqr(hitInfo.dst);
            storeRT(f.Result, px, {col.x, col.y, col.z, 0});
            return;
        }
        Ray ray;
        ray.pos = hitInfo.hitPoint;
        ray.dir = cdir;
        ray.invDir = F3(1 / cdir.x, 1 / cdir.y, 1 / cdir.z);
        RayStats st;
        uint64_t rays = 0;
        TraceInfo traceInfo = Trace(ray, hitInfo, pixelIndex, debugData, state, st, rays);
        float avgCandidateWeight = candidateReservoir.totWeights / candidateReservoir.totCandidates;
        float correctionWeight = avgCandidateWeight / candidateReservoir.chosenWeight;
        traceInfo.indirectLight = traceInfo.indirectLight * correctionWeight;
        f3 combinedLight = traceInfo.directLight + (traceInfo.indirectLight / sqr(traceInfo.totTravelDst));
        storeRT(f.Result, px, {combinedLight.x, combinedLight.y, combinedLight.z, 0});
        if (f.records && rays) f.records[pixelIndex].bounceVisitHash = st.visitHash;
        cnt.rays += rays; cnt.pops += st.pops; cnt.bvChecks += st.bvChecks; cnt.objVisits += st.objVisits;
        cnt.triIters += st.triIters; cnt.triChecks += st.triChecks;
    }
};

void addCounters(orc_counters* dst, const orc_counters& c) {
    if (!dst) return;
    dst->rays += c.rays; dst->pops += c.pops; dst->bvChecks += c.bvChecks; dst->objVisits += c.objVisits;
    dst->triIters += c.triIters; dst->triChecks += c.triChecks; dst->validNeighbours += c.validNeighbours;
    dst->activePixels += c.activePixels;
    dst->temporalCopies += c.temporalCopies; dst->temporalMerges += c.temporalMerges; dst->temporalRejects += c.temporalRejects;
}

// Runs `fn(x, y, counters)` for every pixel of the band, in 8x8 tiles (the reference's thread groups,
// Constants.hlsl:7), tiles distributed dynamically over threads.
template <class Fn>
int forEachPixel(const orc_scene* s, const orc_params* p, orc_frame* f, int threads, Fn fn) {
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
        orc_counters local;
        std::memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 8)
        for (int64_t t = 0; t < ntiles; t++) {
            int ty = ty0 + (int)(t / txn), tx = (int)(t % txn);
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++) {
                    int x = tx * 8 + lx, y = ty * 8 + ly;
                    if (x >= W || y >= H || y < r0 || y >= r1) continue;
                    fn(ctx, (uint32_t)x, (uint32_t)y, local);
                }
        }
#pragma omp critical
        addCounters(&total, local);
    }
    addCounters(f->counters, total);
    return 0;
}

}  // namespace

extern "C" {

int orc_num_threads(void) { return omp_get_max_threads(); }

// PreCalc.compute:11-21,39-50
int orc_calc_tri_normals(orc_scene* s, int threads) {
    if (!s) return -1;
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads)
    for (int64_t i = 0; i < s->nTriangles; i++) {
        rt_triangle tri = s->Triangles[i];
        f3 v0 = F3(ld(s->Vertices, s->nVertices, tri.vertex0Index).pos);
        f3 v1 = F3(ld(s->Vertices, s->nVertices, tri.vertex1Index).pos);
        f3 v2 = F3(ld(s->Vertices, s->nVertices, tri.vertex2Index).pos);
        f3 ab = v1 - v0;
        f3 ac = v2 - v0;
        tri.localNormal = R3(normalize(cross(ab, ac)));
        s->Triangles[i] = tri;
    }
    return 0;
}

int orc_initial_trace(const orc_scene* s, const orc_params* p, orc_frame* f, int threads) {
    return forEachPixel(s, p, f, threads, [](const Ctx& c, uint32_t x, uint32_t y, orc_counters& n) { c.InitialTrace(x, y, n); });
}
int orc_temporal_reuse(const orc_scene* s, const orc_params* p, orc_frame* f, int threads) {
    return forEachPixel(s, p, f, threads, [](const Ctx& c, uint32_t x, uint32_t y, orc_counters& n) { c.TemporalReuse(x, y, n); });
}
int orc_spatial_reuse_pass(const orc_scene* s, const orc_params* p, orc_frame* f, int threads) {
    return forEachPixel(s, p, f, threads, [](const Ctx& c, uint32_t x, uint32_t y, orc_counters& n) { c.SpatialReusePass(x, y, n); });
}
int orc_transfer_to_original(const orc_scene* s, const orc_params* p, orc_frame* f, int threads) {
    return forEachPixel(s, p, f, threads, [](const Ctx& c, uint32_t x, uint32_t y, orc_counters& n) { c.TransferToOriginal(x, y, n); });
}
int orc_trace_rays(const orc_scene* s, const orc_params* p, orc_frame* f, int threads) {
    return forEachPixel(s, p, f, threads, [](const Ctx& c, uint32_t x, uint32_t y, orc_counters& n) { c.TraceRays(x, y, n); });
}

// PostProcessor.compute:10-19.  The kernel has no bounds check; out-of-range threads read 0 and their writes
// are dropped, so only in-range pixels matter.
int orc_accumulate_frames(const orc_params* p, orc_frame* f, int frameCount, int threads) {
    if (!p || !f) return -1;
    int W = p->Resolution[0], H = p->Resolution[1];
    int r0 = 0, r1 = H;
    if (p->row1 > 0) { r0 = p->row0; r1 = p->row1 < H ? p->row1 : H; }
    if (threads <= 0) threads = omp_get_max_threads();
    float weight = 1.0f / (float)(frameCount + 1);
#pragma omp parallel for num_threads(threads)
    for (int64_t i = (int64_t)r0 * W; i < (int64_t)r1 * W; i++) {
        rt_float4 n = f->Result[i], a = f->AccumulatedResult[i];
        rt_float4 o;
        o.x = saturate(a.x * (1 - weight) + n.x * weight);
        o.y = saturate(a.y * (1 - weight) + n.y * weight);
        o.z = saturate(a.z * (1 - weight) + n.z * weight);
        o.w = 0;
        if (p->unorm8Targets) {
            auto q = [](float x) { return (float)(int)(saturate(x) * 255.0f + 0.5f) / 255.0f; };
            o = {q(o.x), q(o.y), q(o.z), 0};
        }
        f->AccumulatedResult[i] = o;
    }
    return 0;
}

uint32_t orc_next_random(uint32_t* state) { return NextRandom(*state); }
float orc_rand_normalized(uint32_t* state) { return randNormalized(*state); }
int32_t orc_rand_int_span(int32_t a, int32_t b, uint32_t* state) { return randIntSpan(a, b, *state); }
int32_t orc_weighted_rand(float a, float b, uint32_t* state) { return weightedRand(a, b, *state) ? 1 : 0; }
float orc_ray_bv_intersect(const float pos[3], const float dir[3], const rt_bv* bv) {
    Ray r;
    r.pos = F3(pos[0], pos[1], pos[2]);
    r.dir = F3(dir[0], dir[1], dir[2]);
    r.invDir = F3(1 / dir[0], 1 / dir[1], 1 / dir[2]);
    return Ctx::RayBVIntersect(r, *bv);
}
int32_t orc_ray_triangle_intersect(const float pos[3], const float dir[3], const rt_vertex* v0, const rt_vertex* v1,
                                   const rt_vertex* v2, float* dst, float uv[2]) {
    rt_vertex verts[3] = {*v0, *v1, *v2};
    orc_scene s;
    std::memset(&s, 0, sizeof s);
    s.Vertices = verts;
    s.nVertices = 3;
    orc_params p;
    std::memset(&p, 0, sizeof p);
    orc_frame f;
    std::memset(&f, 0, sizeof f);
    Ctx c(s, p, f);
    rt_triangle t;
    std::memset(&t, 0, sizeof t);
    t.vertex0Index = 0; t.vertex1Index = 1; t.vertex2Index = 2;
    Ray r;
    r.pos = F3(pos[0], pos[1], pos[2]);
    r.dir = F3(dir[0], dir[1], dir[2]);
    r.invDir = F3(1 / dir[0], 1 / dir[1], 1 / dir[2]);
    f2 o;
    bool hit = c.RayTriangleIntersect(r, t, *dst, o);
    uv[0] = o.x; uv[1] = o.y;
    return hit ? 1 : 0;
}
float orc_get_brdf(const float in[3], const float out[3], const float n[3], float metallicity, float roughness) {
    return GetBRDF(F3(in[0], in[1], in[2]), F3(out[0], out[1], out[2]), F3(n[0], n[1], n[2]), metallicity, roughness);
}
float orc_detmath1(int which, float x) {
    switch (which) {
        case 0: return rtdm::sin_(x);
        case 1: return rtdm::cos_(x);
        case 2: return rtdm::log_(x);
        case 3: return rtdm::exp_(x);
        case 4: return rtdm::asin_(x);
        case 5: return rtdm::acos_(x);
        case 6: return rtdm::atan_(x);
    }
    return 0;
}
float orc_detmath2(int which, float x, float y) { return which == 0 ? rtdm::pow_(x, y) : rtdm::atan2_(x, y); }

}  // extern "C"
