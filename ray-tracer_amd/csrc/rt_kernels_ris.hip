// rt_kernels_ris.hip — gfx950 kernels of the RIS project's ray tracer (BVH_RIS_DL).
//
// What each kernel computes follows BVH_RIS_DL/Assets/Scripts/ComputeShaders/RIS_BVH_RayTracer.compute ("RI:line"),
// RendererResources.hlsl ("RR:line") and PreCalc.compute of that project.  The traversal core, thread-to-pixel
// mapping, LDS stack and counters are the ones of rt_kernels.hip (traverse3 with RIS = true); the reference's
// GenerateCandidates kernel is split into a primary-ray kernel and a candidate kernel, as InitialTrace is.
// Included by rt_native.hip after rt_kernels.hip (one translation unit).
namespace rtn {

DEV bool weightedRandRIS(float a, float b, uint32_t& state) {  // RR:85-92
    float randNorm = randNormalized(state);
    float totalWeight = a + b;
    float relRand = randNorm * totalWeight;
    return relRand < a;
}
DEV v3 lerp3(v3 x, v3 y, float s) { return x + (y - x) * s; }
DEV v3 reflect3(v3 i, v3 n) { return i - (n * 2.0f) * dot(i, n); }
DEV bool anyZero(v3 v) { return v.x == 0 || v.y == 0 || v.z == 0; }  // `all(v) == 0`, RI:463,525

DEV rt_ris_tri loadRisTri(const KParams& P, int i) {
    rt_ris_tri t;
    if (i >= 0 && (int64_t)i < P.nTris) t = P.risTris[i];
    else __builtin_memset(&t, 0, sizeof t);
    return t;
}
DEV rt_ris_material loadRisMaterial(const KParams& P, int i) {
    rt_ris_material m;
    if (i >= 0 && (int64_t)i < P.nMats) m = P.risMats[i];
    else __builtin_memset(&m, 0, sizeof m);
    return m;
}
DEV rt_light_object loadLight(const KParams& P, int i) {
    rt_light_object l;
    if (i >= 0 && (int64_t)i < P.nLights) l = P.lights[i];
    else __builtin_memset(&l, 0, sizeof l);
    return l;
}
DEV float4 loadRisCandidate(const KParams& P, int i) {
    if (i < 0 || (int64_t)i >= P.nRisCand) return make_float4(0, 0, 0, 0);
    return P.risCand[i];
}
DEV void storeRisCandidate(const KParams& P, int i, float4 v) {
    if (i >= 0 && (int64_t)i < P.nRisCand) P.risCand[i] = v;
}

struct RisHit { float dst; v3 hitPoint; v2 uv; v3 normal; int materialIndex; };
// What TraverseSceneBVH leaves in hitInfo (RI:207-214): world hit point, the triangle's stored world normal.
DEV RisHit resolveRisHit(const KParams& P, const Trav& t, v3 wpos, v3 wdir) {
    RisHit h;
    h.dst = t.dst;
    h.hitPoint = V3(-1, -1, -1);
    h.uv = {0, 0};
    h.normal = V3(0, 0, 0);
    h.materialIndex = 0;
    if (t.tri >= 0) {
        const rt_ris_tri tri = P.risTris[t.tri];
        float w = 1 - t.u - t.v;
        h.uv.x = (tri.uvA.x * w + tri.uvB.x * t.u) + tri.uvC.x * t.v;
        h.uv.y = (tri.uvA.y * w + tri.uvB.y * t.u) + tri.uvC.y * t.v;
        h.normal = V3(tri.worldNormal);
        h.hitPoint = wpos + wdir * t.dst;
        h.materialIndex = P.objs[t.obj].materialIndex;
    }
    return h;
}
DEV RisHit loadRisHit(const KParams& P, int64_t i) {
    RisHit h;
    if (i < 0 || i >= P.nRisHit) { h.dst = 0; h.hitPoint = V3(0, 0, 0); h.uv = {0, 0}; h.normal = V3(0, 0, 0); h.materialIndex = 0; return h; }
    const rt_ris_hit_info r = P.risHit[i];
    h.dst = r.dst; h.hitPoint = V3(r.hitPoint); h.uv = {r.uv.x, r.uv.y}; h.normal = V3(r.normal); h.materialIndex = r.materialIndex;
    return h;
}

// =====================================================================================================
// GenerateCandidates, first half (RI:384-412, InitialTrace RI:361-376 without the candidates): camera ray,
// closest hit, HitInfos / RayHitPoints / DebugOverlay.
// =====================================================================================================
template <bool INSTR>
__global__ __launch_bounds__(256, RT_MIN_WAVES) void k_ris_primary(const KParams P) {
    extern __shared__ uint32_t lds[];
    const Pixel px = pixelOfThread(P);
    uint32_t rays = 0;
    Trav t;
    t.dst = INF();
    if (INSTR) { t.pops = t.objVisits = t.triIters = t.iters = 0; t.triChecks = t.bvChecks = 0; }
    travResetWaveCounts<INSTR>(t);
    if (px.active) {
        const uint32_t idx = (uint32_t)px.x, idy = (uint32_t)px.y;
        float focalPosX = ((float)idx / (float)P.ResX - 0.5f) * 2.0f * P.ViewSpaceDimsX;
        float focalPosY = ((float)idy / (float)P.ResY - 0.5f) * 2.0f * P.ViewSpaceDimsY;
        v3 focalPos = V3(focalPosX, focalPosY, 1) * P.FocalPlaneFactor;
        v3 cam = V3(P.camPos[0], P.camPos[1], P.camPos[2]);
        v3 worldFocalPos = mul33(P.camT, focalPos) + cam;
        const uint32_t rayIndex = ((uint32_t)P.ResX * idy + idx) * (uint32_t)P.RaysPerPixel;
        v3 rdir = normalize(worldFocalPos - cam);

        TRAVERSE<INSTR, true>(P, cam, rdir, rdir, laneStack(lds, P.stackDepth), t);
        rays = 1;
        const RisHit h = resolveRisHit(P, t, cam, rdir);
        const bool didHit = h.dst != INF();
        if ((int64_t)rayIndex < P.nRisHit) {
            rt_ris_hit_info o;
            o.dst = h.dst; o.hitPoint = {h.hitPoint.x, h.hitPoint.y, h.hitPoint.z}; o.uv = {h.uv.x, h.uv.y};
            o.normal = {h.normal.x, h.normal.y, h.normal.z}; o.materialIndex = h.materialIndex;
            P.risHit[rayIndex] = o;
        }
        const int64_t i = (int64_t)idy * P.ResX + idx;
        v3 firstHit = didHit ? h.hitPoint : V3(0, 0, 0);
        storeRT(P.rayHitPoints, i, make_float4(firstHit.x, firstHit.y, firstHit.z, 0.0f));
        const bool badMaxBVHDepth = P.risMaxBVHDepth >= (int32_t)MAX_BVH_DEPTH;  // `>=` (RI:410)
        storeRT(P.debugOverlay, i, make_float4(t.triChecks / (float)P.DebugMaxX, t.bvChecks / (float)P.DebugMaxY,
                                               badMaxBVHDepth ? 1.0f : 0.0f, 0.0f));
        if (INSTR && P.records) {
            uint32_t* r = P.records + i * RECORD_DWORDS;
            r[0] = (uint32_t)t.tri; r[1] = (uint32_t)t.obj; r[2] = t.hash; r[3] = 0xffffffffu; r[4] = 0xffffffffu;
            r[5] = 0; r[6] = 0xffffffffu; r[7] = 0;
        }
    }
    writeTileCost(P, px.active && t.dst != INF());
    if (P.stats) {
        waveAdd(P.stats + ST_RAYS, rays);
        if (INSTR) {
            waveAdd(P.stats + ST_ACTIVE, px.active ? 1u : 0u);
            waveAdd(P.stats + ST_POPS, px.active ? t.pops : 0);
            waveAdd(P.stats + ST_BVCHECKS, px.active ? (uint32_t)t.bvChecks : 0);
            waveAdd(P.stats + ST_OBJVISITS, px.active ? t.objVisits : 0);
            waveAdd(P.stats + ST_TRIITERS, px.active ? t.triIters : 0);
            waveAdd(P.stats + ST_TRICHECKS, px.active ? (uint32_t)t.triChecks : 0);
        }
    }
}

// =====================================================================================================
// GenerateCandidates, second half (GenerateCandidates2, RI:326-359): SceneObjectCandidatesNum x TriCandidatesNum
// light samples per hit pixel.  No ray is cast here.  The primary ray draws no random numbers, so the RNG state
// starts from the pixel's seed exactly as it does in the reference's single kernel.
// =====================================================================================================
// GetTriArea (RR:117-124) = length(cross(vB - vA, vC - vA)) * 0.5.  The cross product is the N of the baked triangle,
// so the area is computed once per triangle when the scene is baked (same operations, same bits) and the two or
// more area look-ups per candidate read 4 bytes instead of a 76-byte triangle.
DEV float risTriArea(const KParams& P, int i) {
    if (i < 0 || (int64_t)i >= P.nTris) return 0.0f;  // an all-zero triangle has area 0
    return P.btri[(int64_t)i * 4 + 3].x;
}
DEV int risSceneObjectReservoirSample(const KParams& P, uint32_t& state) {  // RI:273-293
    int firstIndex = randIntSpan(0, P.risEmittingObjectsNum, state);
    int chosenIndex = firstIndex;
    float totWeights = loadLight(P, firstIndex).areaApprox;
    for (int i = 0; i < P.risSceneObjectReservoirTestsNum; i++) {
        int candidateIndex = randIntSpan(0, P.risEmittingObjectsNum, state);
        float candidateWeight = loadLight(P, candidateIndex).areaApprox;
        if (weightedRandRIS(candidateWeight, totWeights, state)) chosenIndex = candidateIndex;
        totWeights += candidateWeight;
    }
    return chosenIndex;
}
DEV int risTriReservoirSample(const KParams& P, int startTriIndex, int endTriIndex, uint32_t& state) {  // RI:295-315
    int firstIndex = randIntSpan(startTriIndex, endTriIndex, state);
    int chosenIndex = firstIndex;
    float totWeights = risTriArea(P, firstIndex);
    for (int i = 0; i < P.risTriReservoirTestsNum; i++) {
        int candidateIndex = randIntSpan(startTriIndex, endTriIndex, state);
        float candidateWeight = risTriArea(P, candidateIndex);
        if (weightedRandRIS(candidateWeight, totWeights, state)) chosenIndex = candidateIndex;
        totWeights += candidateWeight;
    }
    return chosenIndex;
}

// A pixel's candidates are 16 bytes each, Tot of them in a row: written straight from the lane that computes them, a
// wave's 64 stores land 16*Tot bytes apart and every 128-byte line leaves the L2 partly filled several times.  So the
// wave stages 8 candidates per pixel in LDS and then writes them out 8 lanes per pixel: whole 128-byte runs.
constexpr int RIS_STAGE = 8;                   // candidates staged per pixel
constexpr int RIS_STAGE_ROW = RIS_STAGE + 1;   // float4s per staged pixel (one of padding: spreads the LDS banks)
// Dynamic LDS: risStageBytes per wave of the workgroup (four waves, or one: see waveInBlock).
constexpr size_t risStageBytes = 64 * RIS_STAGE_ROW * sizeof(float4) + 64 * sizeof(int);
__global__ __launch_bounds__(256) void k_ris_candidates(const KParams P) {
    extern __shared__ float4 stageAll[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    float4* stage = stageAll + wave * 64 * RIS_STAGE_ROW;
    int* rowBase = (int*)(stageAll + waves * 64 * RIS_STAGE_ROW) + wave * 64;
    const Pixel px = pixelOfThread(P);
    uint32_t generated = 0;
    const uint32_t idx = (uint32_t)px.x, idy = (uint32_t)px.y;
    const uint32_t rayIndex = ((uint32_t)P.ResX * idy + idx) * (uint32_t)P.RaysPerPixel;
    RisHit h;
    bool hit = false;
    if (px.active) {
        h = loadRisHit(P, (int64_t)rayIndex);
        hit = (int64_t)rayIndex < P.nRisHit && h.dst != INF();
    }
    if (hit && P.risTag) P.risTag[rayIndex] = P.risTagValue;   // this copy now holds the pixel's newest candidates
    if (__builtin_amdgcn_ballot_w64(hit) != 0) {  // wave-uniform: the loops below are walked by the whole wave
        uint32_t state = idy * (uint32_t)P.ResX + idx + (uint32_t)P.risFrameRand;
        for (int i = 0; i < P.SceneObjectCandidatesNum; i++) {
            rt_light_object lightObject;
            int startTriIndex = 0, endTriIndex = 0;
            if (hit) {
                const int lightObjectIndex = risSceneObjectReservoirSample(P, state);
                lightObject = loadLight(P, lightObjectIndex);
                startTriIndex = lightObject.triStart;
                endTriIndex = lightObject.triStart + lightObject.totTris;
            }
            for (int j0 = 0; j0 < P.TriCandidatesNum; j0 += RIS_STAGE) {
                const int nj = P.TriCandidatesNum - j0 < RIS_STAGE ? P.TriCandidatesNum - j0 : RIS_STAGE;
                // index of this pixel's candidate j0 of light i, or -1 if the pixel generates none
                rowBase[lane] = hit ? (int)rayIndex * P.TotCandidatesNum + i * P.TriCandidatesNum + j0 : -1;
                if (hit) {
                    for (int jj = 0; jj < nj; jj++) {
                        const int triIndex = risTriReservoirSample(P, startTriIndex, endTriIndex, state);
                        const rt_ris_tri tri = loadRisTri(P, triIndex);
                        // GetRandWorldPointTri, RR:126-143
                        float r1 = randNormalized(state);
                        float r2 = randNormalized(state);
                        if (r1 + r2 > 1.0f) { r1 = 1.0f - r1; r2 = 1.0f - r2; }
                        v3 localPoint = V3(tri.vA) * (1.0f - r1 - r2) + V3(tri.vB) * r1 + V3(tri.vC) * r2;
                        v3 worldPoint = mul44(lightObject.localToWorldMatrix.m, localPoint, 1.0f);
                        v3 dst = worldPoint - h.hitPoint;
                        v3 dir = normalize(dst);
                        generated++;
                        float4 c = make_float4(0, 0, 0, 0);  // a direction below the surface is abandoned (RI:347-352)
                        if (!(dot(h.normal, dir) < 0)) {
                            float hitAngleDot = fmaxf_(dot(V3(tri.worldNormal), dir), 0);
                            float BRDF = 1.0f;
                            float weight = lightObject.brightness * BRDF * hitAngleDot / dot(dst, dst);  // RI:317-323
                            c = make_float4(dir.x, dir.y, dir.z, weight);
                        }
                        stage[lane * RIS_STAGE_ROW + jj] = c;
                    }
                }
                // the wave's lanes run in lockstep and the compiler orders LDS accesses with waitcnt: no barrier needed
                // for data a wave exchanges with itself, only a fence against reordering
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (int r = 0; r < 64; r += 64 / RIS_STAGE) {
                    const int row = r + (int)(lane / RIS_STAGE), piece = (int)(lane % RIS_STAGE);
                    const int base = rowBase[row];
                    if (base >= 0 && piece < nj) storeRisCandidate(P, base + piece, stage[row * RIS_STAGE_ROW + piece]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    if (P.stats && P.statsLevel >= 2) waveAdd(P.stats + ST_NEIGHBOURS, generated);
}

// =====================================================================================================
// TraceRays (RI:503-559; RayDirFromCandidates RI:416-453; Trace RI:455-501)
// =====================================================================================================
// ACC: the launch also makes AccumulateFrames' running average (PostProcesser.compute) of the colour it stores.
template <bool INSTR, bool ACC = false>
__global__ __launch_bounds__(256, RT_MIN_WAVES) void k_ris_trace_rays(const KParams P) {
    extern __shared__ uint32_t lds[];
    const Pixel px = pixelOfThread(P);
    uint32_t rays = 0, pops = 0, bvc = 0, objv = 0, trii = 0, tric = 0;
    float4 colour = make_float4(0, 0, 0, 0);
    if (px.active) {
        const uint32_t idx = (uint32_t)px.x, idy = (uint32_t)px.y;
        uint32_t state = idy * (uint32_t)P.ResX + idx + (uint32_t)P.risFrameRand;
        const uint32_t rayIndex = ((uint32_t)P.ResX * idy + idx) * (uint32_t)P.RaysPerPixel;
        const int64_t i = (int64_t)idy * P.ResX + idx;
        const RisHit first = loadRisHit(P, (int64_t)(idy * (uint32_t)P.ResX + idx));
        // RayDirFromCandidates.  The loop re-reads Candidates[firstIndex] (RI:425), not the index it just drew.
        const int startIndex = (int)(rayIndex * (uint32_t)P.TotCandidatesNum);
        const int endIndex = startIndex + P.TotCandidatesNum;
        const int firstIndex = randIntSpan(startIndex, endIndex, state);
        const float firstWeight = loadRisCandidate(P, firstIndex).w;
        int chosenIndex = firstIndex;
        float totWeights = firstWeight;
        for (int k = 0; k < P.CandidateReservoirTestsNum; k++) {
            int candidateIndex = randIntSpan(startIndex, endIndex, state);
            if (firstWeight == 0) continue;
            if (weightedRandRIS(firstWeight, totWeights, state)) chosenIndex = candidateIndex;
            totWeights += firstWeight;
        }
        const float4 chosen = loadRisCandidate(P, chosenIndex);
        const float risW = totWeights / ((float)P.CandidateReservoirTestsNum * chosen.w);
        const v3 risDir = V3(chosen.x, chosen.y, chosen.z);

        if (first.dst == INF()) {
            colour = make_float4(0, 0, 0, 0);
        } else if (anyZero(risDir)) {
            const rt_ris_material material = loadRisMaterial(P, first.materialIndex);
            v3 col = V3(material.color) * material.brightness / sqr(first.dst);
            colour = make_float4(col.x, col.y, col.z, 0.0f);
        } else {
            const float correctionWeight = risW * (float)P.RaysPerPixel;
            v3 combined = V3(0, 0, 0);
            uint32_t hash = 2166136261u;
            for (int rpp = 0; rpp < P.RaysPerPixel; rpp++) {
                RisHit hitInfo = first;
                v3 rpos = first.hitPoint, rdir = risDir;
                v3 rayColor = V3(1, 1, 1), incomingLight = V3(0, 0, 0);
                for (int bounceCount = 0; bounceCount <= P.risMaxBounceCount; bounceCount++) {
                    if (hitInfo.dst == INF() || anyZero(rdir)) break;
                    const rt_ris_material material = loadRisMaterial(P, hitInfo.materialIndex);
                    rpos = hitInfo.hitPoint - hitInfo.normal * 0.001f;  // offset INTO the surface, as written (RI:468)
                    bool isSpecularBounce = false;
                    if (bounceCount != 0) {
                        isSpecularBounce = P.ScatterProbability < randNormalized(state);
                        v3 scatterDir = normalize(hitInfo.normal + randPointOnUnitSphere(state));
                        v3 specularDir = reflect3(rdir, hitInfo.normal);
                        rdir = normalize(lerp3(scatterDir, specularDir, material.smoothness * (isSpecularBounce ? 1.0f : 0.0f)));
                        if (dot(hitInfo.normal, rdir) < 0) rdir = -rdir;
                    }
                    v3 emittedLight = V3(material.color) * material.brightness;
                    incomingLight = incomingLight + emittedLight * rayColor / sqr(hitInfo.dst);
                    rayColor = rayColor * lerp3(V3(material.color), V3(material.specularColor), isSpecularBounce ? 1.0f : 0.0f);
                    if (bounceCount != P.risMaxBounceCount) {
                        Trav t;
                        travResetWaveCounts<INSTR>(t);
                        TRAVERSE<INSTR, true>(P, rpos, rdir, rdir, laneStack(lds, P.stackDepth), t);
                        rays++;
                        if (INSTR) {
                            pops += t.pops; bvc += (uint32_t)t.bvChecks; objv += t.objVisits; trii += t.triIters;
                            tric += (uint32_t)t.triChecks;
                            hash = rays == 1 ? t.hash : (hash ^ t.hash) * 16777619u;
                        }
                        hitInfo = resolveRisHit(P, t, rpos, rdir);
                    }
                }
                combined = combined + incomingLight;
            }
            combined = combined / correctionWeight;
            colour = make_float4(combined.x, combined.y, combined.z, 0.0f);
            if (INSTR && P.records && rays) P.records[i * RECORD_DWORDS + 7] = hash;
        }
        if (P.result.unorm8) colour = make_float4(toUnorm8(colour.x), toUnorm8(colour.y), toUnorm8(colour.z), toUnorm8(colour.w));
        P.result.rgba[i] = colour;
        if (ACC) {  // k_accumulate on the texel just stored
            const float4 a = P.accumulated.rgba[i];
            const float weight = 1.0f / (float)(P.ppFrameCount + 1);
            float4 o;
            o.x = saturate(a.x * (1 - weight) + colour.x * weight);
            o.y = saturate(a.y * (1 - weight) + colour.y * weight);
            o.z = saturate(a.z * (1 - weight) + colour.z * weight);
            o.w = 0.0f;
            storeRT(P.accumulated, i, o);
        }
    }
    if (P.stats) {
        waveAdd(P.stats + ST_RAYS, rays);
        if (INSTR) {
            waveAdd(P.stats + ST_ACTIVE, px.active ? 1u : 0u);
            waveAdd(P.stats + ST_POPS, pops); waveAdd(P.stats + ST_BVCHECKS, bvc); waveAdd(P.stats + ST_OBJVISITS, objv);
            waveAdd(P.stats + ST_TRIITERS, trii); waveAdd(P.stats + ST_TRICHECKS, tric);
        }
    }
}

// CalcTriNormals of the RIS project (PreCalc.compute:18-28,30-41): world-space normal through the parent object.
__global__ __launch_bounds__(256) void k_ris_calc_tri_normals(rt_ris_tri* tris, int64_t nTris, const rt_scene_object* objs,
                                                              int64_t nObjs, uint32_t trisNum) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)trisNum || i >= nTris) return;
    rt_ris_tri tri = tris[i];
    v3 localNormal = normalize(cross(V3(tri.vB) - V3(tri.vA), V3(tri.vC) - V3(tri.vA)));
    float zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* m = (tri.parentIndex >= 0 && (int64_t)tri.parentIndex < nObjs) ? objs[tri.parentIndex].localToWorldMatrix.m : zero;
    v3 n = normalize(mul44(m, localNormal, 0.0f));
    tris[i].worldNormal = {n.x, n.y, n.z};
}

// Baked triangles (see rt_device.h) from the RIS project's inline-vertex triangles; the fourth quad holds the area.
__global__ void k_ris_bake_tris(const rt_ris_tri* tris, int64_t nTris, float4* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTris) return;
    const rt_ris_tri tri = tris[i];
    v3 v0 = V3(tri.vA);
    v3 eAB = V3(tri.vB) - v0, eAC = V3(tri.vC) - v0;
    v3 N = cross(eAB, eAC);
    out[i * 4 + 0] = make_float4(v0.x, v0.y, v0.z, N.x);
    out[i * 4 + 1] = make_float4(eAB.x, eAB.y, eAB.z, N.y);
    out[i * 4 + 2] = make_float4(eAC.x, eAC.y, eAC.z, N.z);
    out[i * 4 + 3] = make_float4(length(N) * 0.5f, 0, 0, 0);  // GetTriArea, see risTriArea
}

}  // namespace rtn
