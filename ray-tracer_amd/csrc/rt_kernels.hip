// rt_kernels.hip — gfx950 (CDNA4, wave64) kernels of the ReSTIR direct-lighting ray tracer.
//
// What each kernel computes follows BVH_ReSTIR_DL/Assets/Scripts/ComputeShaders/ReSTIR_BVH_RayTracer.compute
// (cited per function as "RS:line"), PreCalc.compute and PostProcessor.compute; how it computes it is specific
// to this GPU:
//   * one wave = one 8x8 pixel tile (coherent primary rays), 4 waves per workgroup, workgroups dealt to the
//     8 XCDs in contiguous runs of tiles so that neighbouring tiles share an L2;
//   * BVH traversal reads one aligned 64-byte record per internal node (both child boxes + both child stack
//     entries) and one aligned 64-byte record per triangle; the per-lane traversal stack lives in LDS,
//     lane-interleaved (conflict-free), one dword per entry, shared by the scene level and the mesh level;
//   * the reference's InitialTrace is split into a primary-ray kernel and a candidate + shadow-ray kernel;
//   * per-pixel records that are only read at the thread's own pixel are planes of float4 (16-byte coalesced
//     accesses); the reservoirs that SpatialReusePass gathers from neighbouring pixels are aligned 64-byte records.
// Arithmetic follows the fp32 conventions documented in include/rt_detmath.h: no contraction
// (-ffp-contract=off), IEEE divide/sqrt, fixed summation order, so results are bit-identical to the CPU oracle.
#include "rt_device.h"

#include "../../include/rt_detmath.h"

namespace rtn {

#define DEV __device__ __forceinline__

constexpr float PI_F = 3.1415926f;        // Constants.hlsl:3
constexpr uint32_t MAX_BVH_DEPTH = 32;    // Constants.hlsl:14
constexpr int WAVES_PER_BLOCK = 4;
#ifndef RT_LEAF_NUM
#define RT_LEAF_NUM 1
#define RT_LEAF_DEN 2
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 1
#endif
#define TRAVERSE traverse3
#ifndef RT_RIS_FAST_DIV
#define RT_RIS_FAST_DIV 1   // 0: the RIS project's box test always takes the division sequence (for A/B measurements)
#endif

struct v2 { float x, y; };
struct v3 { float x, y, z; };
DEV v3 V3(float x, float y, float z) { return {x, y, z}; }
DEV v3 V3(rt_float3 a) { return {a.x, a.y, a.z}; }
DEV v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV v3 operator*(v3 a, v3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
DEV v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
DEV v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }
DEV float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DEV v3 cross(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV float length(v3 a) { return __builtin_sqrtf(dot(a, a)); }
DEV v3 normalize(v3 a) { float inv = 1.0f / __builtin_sqrtf(dot(a, a)); return a * inv; }
// 1.0f / x, correctly rounded, in 3 instructions where that is proven: for 2^-126 <= |x| < 2^126 (exponent field 1..252:
// x and 1/x both normal) v_rcp_f32 followed by one Newton step in fused multiply-adds equals the IEEE quotient for every
// one of those 4.2e9 inputs on gfx950 (tools/micro/rcp_exhaustive.hip checks all 2^32 bit patterns).  Everything else
// (zeros, denormals, huge values, infinities, NaNs) takes the compiler's full division sequence.
DEV float rcpNormalRange(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
DEV float rcpExact(float x) {
    const uint32_t exponent = (__float_as_uint(x) >> 23) & 0xffu;
    if (__builtin_expect(exponent - 1u > 251u, 0)) return 1.0f / x;
    return rcpNormalRange(x);
}
DEV v3 rcp3(v3 a) { return {rcpExact(a.x), rcpExact(a.y), rcpExact(a.z)}; }
// v_min_f32 / v_max_f32 return the non-NaN operand, which is the D3D min/max rule the kernels rely on.
DEV float fminf_(float a, float b) { return __builtin_fminf(a, b); }
DEV float fmaxf_(float a, float b) { return __builtin_fmaxf(a, b); }
DEV float saturate(float x) { return fminf_(fmaxf_(x, 0.0f), 1.0f); }
DEV float clampf(float x, float lo, float hi) { return fminf_(fmaxf_(x, lo), hi); }
DEV float sqr(float a) { return a * a; }
DEV float avg(float a, float b) { return .5f * (a + b); }
DEV float INF() { return __builtin_inff(); }

// mul(float4x4, float4(v, w)).xyz, column-major storage.
DEV v3 mul44(const float* m, v3 v, float w) {
    v3 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * w;
    return r;
}
DEV v3 mul33(const float* m, v3 v) {
    return {(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z,
            (m[6] * v.x + m[7] * v.y) + m[8] * v.z};
}

// ---------------- RNG (RendererResources.hlsl:63-113) ----------------
DEV uint32_t NextRandom(uint32_t& state) {
    state = state * 747796405u + 2891336453u;
    uint32_t result = ((state >> ((state >> 28) + 4)) ^ state) * 277803737u;
    result = (result >> 22) ^ result;
    return result;
}
DEV float randNormalized(uint32_t& state) { return (float)NextRandom(state) / 4294967296.0f; }
DEV int randIntSpan(int a, int b, uint32_t& state) {
    float randNorm = randNormalized(state);
    int diff = b - a;
    int offset = rtdm::f2i((float)diff * randNorm);
    return a + offset;
}
DEV bool weightedRand(float a, float b, uint32_t& state) {
    float randNorm = randNormalized(state);
    float relRand = randNorm * b;
    return relRand < a;
}
DEV float randValueNormalDistribution(uint32_t& state) {
    float theta = 2 * PI_F * randNormalized(state);
    float rho = __builtin_sqrtf(-2 * rtdm::log_(randNormalized(state)));
    return rho * rtdm::cos_(theta);
}
DEV v3 randPointOnUnitSphere(uint32_t& state) {
    float x = randValueNormalDistribution(state);
    float y = randValueNormalDistribution(state);
    float z = randValueNormalDistribution(state);
    return normalize(V3(x, y, z));
}
DEV v2 randPointInCircle(uint32_t& state) {
    float angle = randNormalized(state) * 2 * PI_F;
    float c = rtdm::cos_(angle), s = rtdm::sin_(angle);
    float r = __builtin_sqrtf(randNormalized(state));
    return {c * r, s * r};
}
DEV float angleBetweenNormals(v3 a, v3 b) {
    float d = clampf(dot(a, b), -1.0f, 1.0f);
    return rtdm::acos_(d) * 57.29577951308232f;
}

// ---------------- BRDF.hlsl (x channel; see the comment on GetBRDF's return type at BRDF.hlsl:25,64) ----------------
DEV float GetBRDF(v3 incomingDir, v3 outgoingDir, v3 hitNormal, float metallicity, float roughness) {
    metallicity = 1;
    incomingDir = -incomingDir;
    const float hitAlbedoX = 1.0f;
    v3 H = normalize(incomingDir + outgoingDir);
    float NdotL = saturate(dot(hitNormal, incomingDir));
    float NdotV = saturate(dot(hitNormal, outgoingDir));
    float NdotH = saturate(dot(hitNormal, H));
    float VdotH = saturate(dot(outgoingDir, H));
    float F0x = 0.04f + metallicity * (hitAlbedoX - 0.04f);
    float Fx = F0x + (1.0f - F0x) * rtdm::pow_(1.0f - VdotH, 5.0f);
    float a = roughness * roughness;
    float a2 = a * a;
    float NdotH2 = NdotH * NdotH;
    float denom = (NdotH2 * (a2 - 1.0f) + 1.0f);
    float D = a2 / (3.14f * denom * denom);
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    float G1 = NdotL / (NdotL * (1.0f - k) + k);
    float G2 = NdotV / (NdotV * (1.0f - k) + k);
    float G = G1 * G2;
    float numeratorX = D * G * Fx;
    float denominator = 4.0f * NdotL * NdotV + 0.001f;
    float specularX = numeratorX / denominator;
    float kDx = 1.0f - Fx;
    kDx *= 1.0f - metallicity;
    float diffuseX = kDx * hitAlbedoX / 3.14f;
    return (diffuseX + specularX) * NdotL * 10;
}

// ---------------- robust loads: out-of-range index -> all-zero element (D3D structured-buffer semantics) ----------------
DEV rt_material loadMaterial(const KParams& P, int i) {
    rt_material m;
    if (i >= 0 && (int64_t)i < P.nMats) m = P.mats[i];
    else __builtin_memset(&m, 0, sizeof m);
    return m;
}
DEV rt_triangle loadTriangle(const KParams& P, int64_t i) {
    rt_triangle t;
    if (i >= 0 && i < P.nTris) t = P.tris[i];
    else __builtin_memset(&t, 0, sizeof t);
    return t;
}
DEV rt_vertex loadVertex(const KParams& P, int i) {
    rt_vertex v;
    if (i >= 0 && (int64_t)i < P.nVerts) v = P.verts[i];
    else __builtin_memset(&v, 0, sizeof v);
    return v;
}
DEV v3 texLoad(const Tex& t, uint32_t x, uint32_t y) {
    if (!t.rgba || x >= (uint32_t)t.w || y >= (uint32_t)t.h) return V3(0, 0, 0);
    float4 c = t.rgba[(int64_t)y * t.w + x];
    return V3(c.x, c.y, c.z);
}
// Store to a 3- or 4-channel render target.  The reference's are 8-bit (TextureHelper.cs:144-155); a target created
// as RT_FORMAT_RGBA8_UNORM converts like D3D/Metal do - NaN -> 0, clamp to [0, 1], scale by 255, add 0.5, truncate -
// and keeps the float the texture unit would return on the next load (n / 255).
DEV float toUnorm8(float x) { return (float)rtdm::f2i(saturate(x) * 255.0f + 0.5f) / 255.0f; }
DEV void storeRT(const Tex& t, int64_t i, float4 v) {
    if (t.unorm8) v = make_float4(toUnorm8(v.x), toUnorm8(v.y), toUnorm8(v.z), toUnorm8(v.w));
    t.rgba[i] = v;
}
// RS:94-100
DEV v3 SampleTexture(v2 uv, uint32_t locX, uint32_t locY, uint32_t dimX, uint32_t dimY, const Tex& t) {
    uint32_t tx = locX + rtdm::f2u(uv.x * (float)dimX);
    uint32_t ty = locY + rtdm::f2u(uv.y * (float)dimY);
    return texLoad(t, tx, ty);
}
DEV v3 SampleAtlas(const KParams& P, v2 uv, rt_int2 loc, rt_int2 dims) {
    return SampleTexture(uv, (uint32_t)loc.x, (uint32_t)loc.y, (uint32_t)dims.x, (uint32_t)dims.y, P.atlas);
}
DEV float matMetallicity(const KParams& P, const rt_material& m, v2 uv) {
    return m.metallicity == -1 ? SampleAtlas(P, uv, m.metallicityTexLoc, m.metallicityTexDims).x : m.metallicity;
}
DEV float matRoughness(const KParams& P, const rt_material& m, v2 uv) {
    return m.roughness == -1 ? SampleAtlas(P, uv, m.roughnessTexLoc, m.roughnessTexDims).x : m.roughness;
}
DEV v3 matCol(const KParams& P, const rt_material& m, v2 uv) {
    bool tex = m.col.x == -1 && m.col.y == -1 && m.col.z == -1;
    return tex ? SampleAtlas(P, uv, m.colTexLoc, m.colTexDims) : V3(m.col);
}

// ---------------- thread -> pixel mapping ----------------
// One wave per 8x8 tile, four waves (a 16x16-pixel block) per workgroup.  Which block a workgroup renders comes
// from a table built on the host (rt_native.hip: buildBlockMap), which orders blocks so that the workgroups that
// land on one XCD (launch index mod 8, the observed round-robin placement; used for speed only) render compact
// image regions, and the regions are dealt evenly to the XCDs.
// A launch may also give every wave a workgroup of its own (64 threads, four times the workgroups): workgroup i then
// renders tile (i / 8) mod 4 of table entry (i / 32) * 8 + i mod 8, so that a block's four tiles stay on one XCD.
struct Pixel { int x, y; bool active; };
DEV uint32_t waveInBlock() { return blockDim.x == 64 ? (blockIdx.x >> 3) & 3u : threadIdx.x >> 6; }
DEV uint32_t tableEntryOfGroup() { return blockDim.x == 64 ? ((blockIdx.x >> 5) << 3) | (blockIdx.x & 7u) : blockIdx.x; }
DEV Pixel pixelOfThread(const KParams& P, uint32_t launchBlock) {
    const uint32_t m = P.blockMap[launchBlock];
    const uint32_t bx = m & 0xffffu, by = m >> 16;
    const uint32_t wave = waveInBlock(), lane = threadIdx.x & 63;
    Pixel p;
    p.x = (int)(bx * 16 + (wave & 1) * 8 + (lane & 7));
    p.y = (P.row0 & ~7) + (int)(by * 16 + (wave >> 1) * 8 + (lane >> 3));
    p.active = m != 0xffffffffu && p.x < P.ResX && p.y < P.ResY && p.y >= P.row0 && p.y < P.row1;
    return p;
}

// Cost estimate of this wave's tile for the launches that follow the primary-ray kernel: its number of hit pixels.
DEV Pixel pixelOfThread(const KParams& P) { return pixelOfThread(P, tableEntryOfGroup()); }
DEV void writeTileCost(const KParams& P, bool hit, uint32_t launchBlock) {
    if (!P.blockCost) return;
    const unsigned long long hitMask = __builtin_amdgcn_ballot_w64(hit);
    const uint32_t m = P.blockMap[launchBlock];
    if ((threadIdx.x & 63) == 0 && m != 0xffffffffu)
        P.blockCost[((size_t)(m >> 16) * P.blocksX + (m & 0xffffu)) * 4 + waveInBlock()] = (uint32_t)__builtin_popcountll(hitMask);
}

// Measured cost of this wave's tile: the steps the wave needed for its secondary rays (0 in lanes that cast none).  The
// regions of the image are ranked by it for the launches of the following frames (k_order_blocks).
DEV void writeTileCost(const KParams& P, bool hit) { writeTileCost(P, hit, tableEntryOfGroup()); }
DEV void writeTileSteps(const KParams& P, uint32_t steps) {
    if (!P.blockSteps) return;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(steps, off, 64); steps = o > steps ? o : steps; }
    const uint32_t m = P.blockMap[tableEntryOfGroup()];
    if ((threadIdx.x & 63) == 0 && m != 0xffffffffu)
        P.blockSteps[((size_t)(m >> 16) * P.blocksX + (m & 0xffffu)) * 4 + waveInBlock()] = steps;
}

// ---------------- BVH traversal ----------------
struct Trav {
    float dst;        // closest hit distance along the world ray, INF on a miss
    float u, v;       // barycentrics of the closest hit
    int tri, obj;     // closest-hit triangle and scene object, -1 on a miss
    uint32_t steps;     // steps the WAVE made while this lane's ray was traced (internal-node steps + trips); a cost measure
    int triChecks, bvChecks;  // debugData of RS:184,202,265,267
    // instrumentation (INSTR only)
    uint32_t hash;
    uint32_t pops, objVisits, triIters, iters;
    // wave-level diagnostics (INSTR only; the same value in every lane): internal-node steps and leaf steps the wave
    // executed, the lanes active in them, and the service steps
    uint32_t wInt, wIntLanes, wLeaf, wLeafLanes, wSvc;
};

DEV float RayBVIntersect(v3 pos, v3 inv, v3 bmin, v3 bmax) {  // RS:148-161
    v3 tMin = (bmin - pos) * inv;
    v3 tMax = (bmax - pos) * inv;
    v3 t1 = {fminf_(tMin.x, tMax.x), fminf_(tMin.y, tMax.y), fminf_(tMin.z, tMax.z)};
    v3 t2 = {fmaxf_(tMin.x, tMax.x), fmaxf_(tMin.y, tMax.y), fmaxf_(tMin.z, tMax.z)};
    float tNear = fmaxf_(fmaxf_(t1.x, t1.y), t1.z);
    float tFar = fminf_(fminf_(t2.x, t2.y), t2.z);
    bool didHit = tFar >= tNear && tFar > 0;
    return didHit ? (tNear > 0 ? tNear : 0.0f) : INF();
}

// Both child boxes of a wide record at once, on register pairs.  The record keeps the x and y bounds of a box together
// and the z bounds of the two boxes together (rt_device.h), so with the ray's (x, y) as one register pair the twelve
// subtractions and twelve multiplications of two RayBVIntersect calls are six v_pk_add_f32 and six v_pk_mul_f32 (the z
// component is broadcast to both halves by the instruction's operand select).  Each component is the same IEEE subtract
// and multiply as in RayBVIntersect (no contraction), so the distances are bit-identical.
typedef float f2v __attribute__((ext_vector_type(2)));
DEV float boxEntry(float tMinX, float tMinY, float tMinZ, float tMaxX, float tMaxY, float tMaxZ) {
    float t1x = fminf_(tMinX, tMaxX), t1y = fminf_(tMinY, tMaxY), t1z = fminf_(tMinZ, tMaxZ);
    float t2x = fmaxf_(tMinX, tMaxX), t2y = fmaxf_(tMinY, tMaxY), t2z = fmaxf_(tMinZ, tMaxZ);
    float tNear = fmaxf_(fmaxf_(t1x, t1y), t1z);
    float tFar = fminf_(fminf_(t2x, t2y), t2z);
    bool didHit = tFar >= tNear && tFar > 0;
    return didHit ? (tNear > 0 ? tNear : 0.0f) : INF();
}
DEV void RayBVIntersectPair(v3 pos, v3 inv, float4 q0, float4 q1, float4 q2, float& dstA, float& dstB) {
    const f2v pxy = {pos.x, pos.y}, pzz = {pos.z, pos.z}, ixy = {inv.x, inv.y}, izz = {inv.z, inv.z};
    const f2v aMin = ((f2v){q0.x, q0.y} - pxy) * ixy, aMax = ((f2v){q0.z, q0.w} - pxy) * ixy;
    const f2v bMin = ((f2v){q1.x, q1.y} - pxy) * ixy, bMax = ((f2v){q1.z, q1.w} - pxy) * ixy;
    const f2v zMin = ((f2v){q2.x, q2.y} - pzz) * izz, zMax = ((f2v){q2.z, q2.w} - pzz) * izz;  // (box A, box B)
    dstA = boxEntry(aMin.x, aMin.y, zMin.x, aMax.x, aMax.y, zMax.x);
    dstB = boxEntry(bMin.x, bMin.y, zMin.y, bMax.x, bMax.y, zMax.y);
}

// The RIS project's variant (BVH_RIS_DL .../RIS_BVH_RayTracer.compute:111-124) divides by the direction instead of
// multiplying by a stored reciprocal; the two round differently, so it is kept as written.
DEV float RayBVIntersectDiv(v3 pos, v3 dir, v3 bmin, v3 bmax) {
    v3 a = bmin - pos, b = bmax - pos;
    v3 tMin = {a.x / dir.x, a.y / dir.y, a.z / dir.z};
    v3 tMax = {b.x / dir.x, b.y / dir.y, b.z / dir.z};
    v3 t1 = {fminf_(tMin.x, tMax.x), fminf_(tMin.y, tMax.y), fminf_(tMin.z, tMax.z)};
    v3 t2 = {fmaxf_(tMin.x, tMax.x), fmaxf_(tMin.y, tMax.y), fmaxf_(tMin.z, tMax.z)};
    float tNear = fmaxf_(fmaxf_(t1.x, t1.y), t1.z);
    float tFar = fminf_(fminf_(t2.x, t2.y), t2.z);
    bool didHit = tFar >= tNear && tFar > 0;
    return didHit ? (tNear > 0 ? tNear : 0.0f) : INF();
}

// The same twelve quotients without the division sequence (ten instructions each), where that is exact.  With
// r = RN(1 / d) (rcpExact), q0 = RN(a * r) lies within one unit in the last place of a / d, the residual e = a - d * q0 is
// exact in one fused multiply-add, and RN(q0 + e * r) is then the correctly rounded quotient (Markstein's correction
// step) - as long as nothing on the way leaves the normal range.  That is secured by ranges, not assumed: the ray's
// direction components within 2^-40 .. 2^40 (checked once per ray and object: divisorsInRange), every numerator of the
// step within 2^-60 .. 2^59 in magnitude (one min / max chain over the twelve); then quotients stay within
// 2^-100 .. 2^99 and residuals above 2^-124.  A step that fails either test - a ray parallel to an axis, an origin on a
// box plane, infinities - takes RayBVIntersectDiv as written.  tools/micro/div_check.hip compares the three-instruction
// quotient with the division over 2^32 random pairs and the edges of those ranges on the GPU.
DEV bool divisorsInRange(v3 d) {
    const uint32_t ex = (__float_as_uint(d.x) >> 23) & 0xffu, ey = (__float_as_uint(d.y) >> 23) & 0xffu, ez = (__float_as_uint(d.z) >> 23) & 0xffu;
    return ex - 87u <= 80u && ey - 87u <= 80u && ez - 87u <= 80u;
}
DEV float quotientInRange(float a, float d, float r) {
    const float q0 = a * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q0, a), r, q0);
}
DEV bool RayBVIntersectDivPairFast(v3 pos, v3 dir, v3 rcp, float4 q0, float4 q1, float4 q2, float& dstA, float& dstB) {
    const float aMinX = q0.x - pos.x, aMinY = q0.y - pos.y, aMinZ = q2.x - pos.z, aMaxX = q0.z - pos.x, aMaxY = q0.w - pos.y, aMaxZ = q2.z - pos.z;
    const float bMinX = q1.x - pos.x, bMinY = q1.y - pos.y, bMinZ = q2.y - pos.z, bMaxX = q1.z - pos.x, bMaxY = q1.w - pos.y, bMaxZ = q2.w - pos.z;
    auto mag = [](float x) { return __builtin_fabsf(x); };
    const float lo = fminf_(fminf_(fminf_(fminf_(mag(aMinX), mag(aMinY)), fminf_(mag(aMinZ), mag(aMaxX))), fminf_(fminf_(mag(aMaxY), mag(aMaxZ)), fminf_(mag(bMinX), mag(bMinY)))),
                            fminf_(fminf_(mag(bMinZ), mag(bMaxX)), fminf_(mag(bMaxY), mag(bMaxZ))));
    const float hi = fmaxf_(fmaxf_(fmaxf_(fmaxf_(mag(aMinX), mag(aMinY)), fmaxf_(mag(aMinZ), mag(aMaxX))), fmaxf_(fmaxf_(mag(aMaxY), mag(aMaxZ)), fmaxf_(mag(bMinX), mag(bMinY)))),
                            fmaxf_(fmaxf_(mag(bMinZ), mag(bMaxX)), fmaxf_(mag(bMaxY), mag(bMaxZ))));
    // (a NaN numerator is dropped by min / max and slips through: its quotient is a NaN either way, and boxEntry's own
    // min / max drop it again)
    if (!(lo >= 0x1p-60f && hi <= 0x1p59f)) return false;
    dstA = boxEntry(quotientInRange(aMinX, dir.x, rcp.x), quotientInRange(aMinY, dir.y, rcp.y), quotientInRange(aMinZ, dir.z, rcp.z),
                    quotientInRange(aMaxX, dir.x, rcp.x), quotientInRange(aMaxY, dir.y, rcp.y), quotientInRange(aMaxZ, dir.z, rcp.z));
    dstB = boxEntry(quotientInRange(bMinX, dir.x, rcp.x), quotientInRange(bMinY, dir.y, rcp.y), quotientInRange(bMinZ, dir.z, rcp.z),
                    quotientInRange(bMaxX, dir.x, rcp.x), quotientInRange(bMaxY, dir.y, rcp.y), quotientInRange(bMaxZ, dir.z, rcp.z));
    return true;
}

// GetRayCollision (RS:284-291) = TraverseSceneBVH (RS:219-282) + TraverseTrisBVH (RS:163-217), as one loop over one
// stack.  `stk` points at this lane's column of the LDS stack (entries 64 dwords apart).  Visit order, the near-first
// rule and the strict `<` culls are the reference's; only what is fetched per step, and when, differs.  Shading data
// of the winning triangle (uv, normals, hit point) is derived afterwards from (tri, obj, u, v).
//
// Organisation for 64-lane execution:
//   * the entry to visit next stays in a register (`e`); only the FAR child of an internal node goes to the LDS stack
//     (the reference pushes far then near and pops near at once: same order, half the stack traffic);
//   * each outer trip runs an internal-node phase - every lane that holds an internal entry steps down the tree until
//     at least half of the lanes still at work are waiting - and then a leaf phase in which the waiting lanes test
//     their triangles together; in a single mixed loop nearly every trip pays for both kinds of work;
//   * leaving an object (RS:251-258), entering the next one (RS:235-246: two matrix products, three IEEE divisions)
//     and ending the ray are a third, "service" step: a lane that needs it holds ENTRY_NONE and waits.  A ray needs
//     it a handful of times, but with 64 lanes at different places almost every trip had some lane on that path.
// RIS = the RIS project's traversal (BVH_RIS_DL .../RIS_BVH_RayTracer.compute:126-238): same loops, but the box test
// divides by the direction and the leaf loop has no back-face pre-cull.
constexpr uint32_t ENTRY_NONE = 0xffffffffu;  // no entry held (never a valid entry: an inline leaf with count 31 at
                                              // start 2^26-1 is rejected when the scene is validated)
constexpr uint32_t ENTRY_DEAD = 0xfffffffeu;  // the lane's ray has ended (likewise never a valid entry: count 31 at start 2^26-2)
constexpr uint32_t LEAF_BITS = ENTRY_LEAF | ENTRY_LEAF_INDIRECT;
// What a lane does next is read off its entry alone, one comparison each: internal entries are below the two leaf bits,
// leaf entries from there up to the two reserved values.
DEV bool entryIsInternal(uint32_t e) { return e < ENTRY_LEAF_INDIRECT; }
DEV bool entryIsLeaf(uint32_t e) { return e - ENTRY_LEAF_INDIRECT < ENTRY_DEAD - ENTRY_LEAF_INDIRECT; }
struct TravState {
    v3 wpos, wdir, winv;     // the ray in world space
    v3 rpos, rdir, rinv;     // ray of the level being traversed (world at scene level, object space inside an object)
    float best, worldBest;   // hitInfo.dst of the level being traversed / of the world level
    uint32_t* sp;            // slot the next push writes (LDS address: a push or pop is one add, no index arithmetic)
    uint32_t* floor;         // slot below which the entries belong to the enclosing level (scene level: the column's start)
    int inObj, objCur, objEnd, curObj, localTri;   // inObj: inside a scene object's BVH
    float localU, localV;
    uint32_t e;              // entry to visit next, or ENTRY_NONE
    uint32_t top;            // newest of the `sp` stacked entries (see travPush)
    int divOk, wDivOk;       // RIS only: the direction of the level being traversed / of the world ray passes divisorsInRange
    const float4* topLds;    // TOP only: this wave's copy of the table of top nodes (see ENTRY_TOP)
};

template <bool INSTR, bool RIS = false, bool TOP = false>
DEV void travBegin(const KParams& P, TravState& s, Trav& t, v3 wpos, v3 wdir, v3 winv, uint32_t* stk) {
    if (RIS) { winv = rcp3(wdir); s.wDivOk = divisorsInRange(wdir) ? 1 : 0; s.divOk = s.wDivOk; }   // (the RIS callers pass no reciprocal)
    s.wpos = wpos; s.wdir = wdir; s.winv = winv;
    s.rpos = wpos; s.rdir = wdir; s.rinv = winv;
    s.best = INF(); s.worldBest = INF();
    s.sp = stk; s.floor = stk; s.inObj = 0; s.objCur = 0; s.objEnd = 0; s.curObj = -1; s.localTri = -1;
    s.localU = 0; s.localV = 0;
    s.e = TOP ? P.sceneRootEntryTop : P.sceneRootEntry;
    s.top = ENTRY_NONE;
    t.tri = -1; t.obj = -1; t.steps = 0; t.u = 0; t.v = 0; t.triChecks = 0; t.bvChecks = 0;
    if (INSTR) { t.hash = 2166136261u; t.pops = 1; t.objVisits = 0; t.triIters = 0; t.iters = 0; }
}
template <bool INSTR>
DEV void travResetWaveCounts(Trav& t) { if (INSTR) t.wInt = t.wIntLanes = t.wLeaf = t.wLeafLanes = t.wSvc = 0; }
// The stack keeps its newest entry in a register: a pop hands that register over at once and refills it from LDS behind
// the fetch of the popped node, instead of putting an LDS read in front of it.  Entry j (counted from the oldest) moves
// to slot j + 1 when the next one is pushed; slot 0 only ever receives the unused register of an empty stack.  The stack
// holds the deepest path of the validated scene plus that slot: there is no overflow to guard against.
DEV void travPush(TravState& s, uint32_t entry) {
    *s.sp = s.top;
    s.top = entry;
    s.sp += 64;
}
DEV void travTake(TravState& s) {  // the stack is not empty
    s.e = s.top;
    s.sp -= 64;
    __builtin_amdgcn_sched_barrier(0);  // (the refill below must not be scheduled above the hand-over: it would need a copy and a wait)
    s.top = *s.sp;
}
// Next entry of the level being traversed, or ENTRY_NONE when that level's part of the stack is used up.
template <bool INSTR>
DEV void travPop(const KParams& P, TravState& s, Trav& t) {
    if (s.sp > s.floor) {
        travTake(s);
        if (INSTR) t.pops++;
    } else {
        s.e = ENTRY_NONE;
    }
}
// Service step of one lane that holds ENTRY_NONE.  Leaves ENTRY_DEAD when the ray is finished.
template <bool INSTR, bool RIS, bool TOP = false>
DEV void travService(const KParams& P, TravState& s, Trav& t, const uint32_t* stk) {
    if (s.inObj) {  // the object's subtree is exhausted (RS:251-258)
        if (s.best < s.worldBest) { s.worldBest = s.best; t.tri = s.localTri; t.obj = s.curObj; t.u = s.localU; t.v = s.localV; }
        s.best = s.worldBest;
        s.rpos = s.wpos; s.rdir = s.wdir; s.rinv = s.winv;
        if (RIS) s.divOk = s.wDivOk;
        s.inObj = 0;
        s.floor = (uint32_t*)stk;
    }
    if (s.objCur < s.objEnd) {  // next scene object of the current scene-level leaf (RS:235-246)
        const float* m = P.objs[s.objCur].worldToLocalMatrix.m;
        s.rpos = mul44(m, s.wpos, 1.0f);
        s.rdir = mul44(m, s.wdir, 0.0f);
        s.rinv = rcp3(s.rdir);
        if (RIS) s.divOk = divisorsInRange(s.rdir) ? 1 : 0;
        s.best = INF();
        s.localTri = -1;
        s.curObj = s.objCur;
        s.inObj = 1;
        s.floor = s.sp;
        s.e = (TOP ? P.objRootEntryTop : P.objRootEntry)[s.objCur];
        s.objCur++;
        if (INSTR) { t.objVisits++; t.pops++; }
        return;
    }
    if (s.sp == stk) { t.dst = s.worldBest; s.e = ENTRY_DEAD; return; }
    travTake(s);
    if (INSTR) t.pops++;
}
// One internal-node step of one lane (RS:195-214).
template <bool INSTR, bool RIS, bool TOP = false>
DEV void travInternal(const KParams& P, TravState& s, Trav& t, uint32_t* stk) {
    static_assert(!(INSTR && TOP), "the visit hash is taken over BV indices: the instrumented kernels do not use the table of top nodes");
    const uint32_t e = s.e;
    if (INSTR) { t.iters++; t.hash = (t.hash ^ (0x80000000u | e)) * 16777619u; }
    float4 q0, q1, q2;
    float2 q3;   // the two child entries
    if (TOP && (e & ENTRY_TOP)) {   // a slot of the wave's table in LDS
        const float4* rec = s.topLds + ((e & (uint32_t)(TOP_MAX - 1)) << 2);
        q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
        q3 = *(const float2*)(rec + 3);
    } else {
        // 32-bit byte offset from a uniform base (one shift instead of 64-bit address arithmetic; the scene check keeps
        // the node and triangle counts below 2^26)
        const float4* rec = (const float4*)((const char*)P.wide + (e << 6));
        q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
        q3 = *(const float2*)(rec + 3);
    }
    float dstA, dstB;
    if (RIS) {
        if (!RT_RIS_FAST_DIV || !s.divOk || !RayBVIntersectDivPairFast(s.rpos, s.rdir, s.rinv, q0, q1, q2, dstA, dstB)) {
            dstA = RayBVIntersectDiv(s.rpos, s.rdir, V3(q0.x, q0.y, q2.x), V3(q0.z, q0.w, q2.z));
            dstB = RayBVIntersectDiv(s.rpos, s.rdir, V3(q1.x, q1.y, q2.y), V3(q1.z, q1.w, q2.w));
        }
    } else {
        RayBVIntersectPair(s.rpos, s.rinv, q0, q1, q2, dstA, dstB);
    }
    t.bvChecks += 2;
    const uint32_t eA = __float_as_uint(q3.x), eB = __float_as_uint(q3.y);
    const bool aFirst = dstA < dstB;  // the nearer child is visited first
    const uint32_t eNear = aFirst ? eA : eB, eFar = aFirst ? eB : eA;
    const bool nearHit = (aFirst ? dstA : dstB) < s.best, farHit = (aFirst ? dstB : dstA) < s.best;
    // Written flat (one predicated push, one predicated pop) instead of as nested branches: fewer exec-mask
    // regions per step.  A far child can pass while the near one fails only when a distance is NaN.
    if (nearHit && farHit) travPush(s, eFar);
    s.e = nearHit ? eNear : eFar;
    if (nearHit || farHit) { if (INSTR) t.pops++; }
    else travPop<INSTR>(P, s, t);
}
// Leaf step of one lane: a run of scene objects at scene level, a run of triangles inside an object (RS:179-193).
template <bool INSTR, bool RIS>
DEV void travLeaf(const KParams& P, TravState& s, Trav& t, const uint32_t* stk) {
    const uint32_t e = s.e;
    int start, count;
    if (e & ENTRY_LEAF) {
        start = (int)(e & (uint32_t)ENTRY_INLINE_MAX_START);
        count = (int)((e >> 26) & 31u);
        if (INSTR) { t.hash = (t.hash ^ (uint32_t)start) * 16777619u; t.hash = (t.hash ^ (uint32_t)count) * 16777619u; }
    } else {
        const rt_bv* bv = &P.BVs[e & ENTRY_INDEX_MASK];
        int a = bv->indexA, b = bv->indexB;
        if (INSTR) { t.hash = (t.hash ^ (uint32_t)a) * 16777619u; t.hash = (t.hash ^ (uint32_t)b) * 16777619u; }
        start = a < 0 ? -a : a;
        count = b < 0 ? -b : b;
    }
    if (INSTR) t.iters++;
    if (!s.inObj) {  // scene-level leaf: its objects are entered one by one by the service step
        s.objCur = start;
        s.objEnd = start + count;
        s.e = ENTRY_NONE;
        return;
    }
    for (int triIndex = start; triIndex < start + count; triIndex++) {
        const float4* rec = (const float4*)((const char*)P.btri + ((uint32_t)triIndex << 6));
        if (INSTR) t.triIters++;
        bool facing = true;
        // the whole 64-byte record instead of a triangle and three dependent vertex gathers.  (The compiled loop tests the
        // back face on the record's last three quarters and fetches the first - v0 - only for a triangle that faces the
        // ray; with all four loads held in one batch the kernels gain under 1 %, within the spread of a run.)
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        if (!RIS) { const float4 q3 = rec[3]; facing = dot(V3(q3.x, q3.y, q3.z), s.rdir) <= 0; }
        if (facing) {
            // RayTriangleIntersect, RS:119-146
            v3 edgeAB = V3(q1.x, q1.y, q1.z), edgeAC = V3(q2.x, q2.y, q2.z);
            v3 normalVector = V3(q0.w, q1.w, q2.w);
            v3 ao = s.rpos - V3(q0.x, q0.y, q0.z);
            v3 dao = cross(ao, s.rdir);
            float determinant = -dot(s.rdir, normalVector);
            // the quotient only matters when determinant >= 1E-8 (didHit below); below 2^126 that is rcpNormalRange's domain
            float invDet = determinant < 0x1p126f ? rcpNormalRange(determinant) : 1 / determinant;
            float dst = dot(ao, normalVector) * invDet;
            float u = dot(edgeAC, dao) * invDet;
            float v = -dot(edgeAB, dao) * invDet;
            float w = 1 - u - v;
            bool didHit = determinant >= 1E-8f && dst >= 0 && u >= 0 && v >= 0 && w >= 0;
            t.triChecks++;
            if (didHit && dst < s.best) { s.best = dst; s.localU = u; s.localV = v; s.localTri = triIndex; }
        }
    }
    travPop<INSTR>(P, s, t);
}
// One outer trip: service, internal-node phase, leaf phase.  A lane whose ray has ended holds ENTRY_DEAD (t.dst is set).
template <bool INSTR, bool RIS, bool TOP = false>
DEV void travTrip(const KParams& P, TravState& s, Trav& t, uint32_t* stk) {
    if (INSTR && __builtin_amdgcn_ballot_w64(s.e == ENTRY_NONE) != 0) t.wSvc++;
    if (s.e == ENTRY_NONE) travService<INSTR, RIS, TOP>(P, s, t, stk);
    // The phase is left once at least RT_LEAF_NUM / RT_LEAF_DEN of the lanes still at work wait for a leaf or for
    // service, i.e. once the lanes holding an internal entry are few enough (the live lanes do not change in it).
    const int nAlive = __builtin_popcountll(__builtin_amdgcn_ballot_w64(s.e != ENTRY_DEAD));
    int nInt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(entryIsInternal(s.e)));
    t.steps++;
    if (nInt != 0) {
        do {
            t.steps++;
            if (INSTR) { t.wInt++; t.wIntLanes += (uint32_t)nInt; }
            if (entryIsInternal(s.e)) travInternal<INSTR, RIS, TOP>(P, s, t, stk);
            nInt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(entryIsInternal(s.e)));
        } while (RT_LEAF_DEN * nInt > (RT_LEAF_DEN - RT_LEAF_NUM) * nAlive);   // (false at the latest when no lane is left)
    }
    if (INSTR) {
        const int nLeaf = __builtin_popcountll(__builtin_amdgcn_ballot_w64(entryIsLeaf(s.e)));
        if (nLeaf) { t.wLeaf++; t.wLeafLanes += (uint32_t)nLeaf; }
    }
    if (entryIsLeaf(s.e)) travLeaf<INSTR, RIS>(P, s, t, stk);
}

template <bool INSTR, bool RIS = false, bool TOP = false>
DEV void traverse3(const KParams& P, const v3 wpos, const v3 wdir, const v3 winv, uint32_t* stk, Trav& t, const float4* topLds = nullptr) {
    TravState s;
    travBegin<INSTR, RIS, TOP>(P, s, t, wpos, wdir, winv, stk);
    s.topLds = topLds;
    while (__builtin_amdgcn_ballot_w64(s.e != ENTRY_DEAD) != 0) travTrip<INSTR, RIS, TOP>(P, s, t, stk);
}
// TOP kernels: the wave's own copy of the table of top nodes, behind the stacks of the workgroup's waves in dynamic LDS
// (P.topCount records of 64 bytes per wave).  Wave-private, so a wave-level fence is all the ordering it needs.
DEV const float4* loadTopTable(const KParams& P, uint32_t* lds) {
    const uint32_t waves = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4* dst = (float4*)(lds + (size_t)waves * P.stackDepth * 64) + (size_t)wave * P.topCount * 4;
    for (int i = (int)lane; i < P.topCount * 4; i += 64) dst[i] = P.top[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return dst;
}

// Shading data of a traversal result, as TraverseSceneBVH leaves it in hitInfo (RS:248-258).
struct HitInfo { float dst; v3 hitPoint; v2 uv; v3 normal; v3 incomingDir; int materialIndex; };
DEV HitInfo resolveHit(const KParams& P, const Trav& t, v3 wpos, v3 wdir) {
    HitInfo h;
    h.dst = t.dst;
    h.hitPoint = V3(-1, -1, -1);
    h.uv = {0, 0};
    h.normal = V3(0, 0, 0);
    h.incomingDir = V3(0, 0, 0);
    h.materialIndex = 0;
    if (t.tri >= 0) {
        const rt_scene_object* o = &P.objs[t.obj];
        const rt_triangle tri = P.tris[t.tri];
        const rt_float2 uv0 = P.verts[tri.vertex0Index].uv, uv1 = P.verts[tri.vertex1Index].uv,
                        uv2 = P.verts[tri.vertex2Index].uv;
        float w = 1 - t.u - t.v;
        h.uv.x = (uv0.x * w + uv1.x * t.u) + uv2.x * t.v;
        h.uv.y = (uv0.y * w + uv1.y * t.u) + uv2.y * t.v;
        v3 n = V3(tri.localNormal);
        const int matIndex = o->materialIndex;
        rt_material mat = loadMaterial(P, matIndex);
        if (mat.normalsTexLoc.x != -1 && mat.normalsTexLoc.y != -1)
            n = SampleAtlas(P, h.uv, mat.normalsTexLoc, mat.normalsTexDims) * 2 - V3(1, 1, 1);
        h.normal = normalize(mul44(o->localToWorldMatrix.m, n, 0.0f));
        h.hitPoint = wpos + wdir * t.dst;
        h.materialIndex = matIndex;
    }
    return h;
}

// ---------------- wave-level counters ----------------
// Counters are kept in STAT_SHARDS copies, 8 counters each; a workgroup adds into the copy chosen by its index, so
// the atomics of one launch spread over many addresses instead of serialising on one (a single contended word
// retires only ~one atomic per 12 ns).  The host sums the copies.
constexpr int STAT_SHARDS = 64;
DEV void waveAdd(unsigned long long* dst, uint32_t v) {
    // Sum over the 64 lanes (callers pass 0 for lanes that have nothing to add).  The always-on ray counter adds 0..7
    // per lane: three ballots give its sum; anything larger takes the butterfly (cross-lane moves through LDS).
    if (__builtin_amdgcn_ballot_w64(v != 0) == 0) return;
    if (__builtin_amdgcn_ballot_w64(v > 7u) == 0) {
        v = (uint32_t)(__builtin_popcountll(__builtin_amdgcn_ballot_w64((v & 1u) != 0)) +
                       2 * __builtin_popcountll(__builtin_amdgcn_ballot_w64((v & 2u) != 0)) +
                       4 * __builtin_popcountll(__builtin_amdgcn_ballot_w64((v & 4u) != 0)));
    } else {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(dst + (blockIdx.x % STAT_SHARDS) * 8, (unsigned long long)v);
}
// Diagnostics (instrumented variants only): lifetime and loop-trip counts of each wave of a launch.
DEV void writeWaveProfile(const KParams& P, unsigned long long tStart, uint32_t iters, const Trav& t) {
    uint32_t mx = iters, sm = iters;
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; sm += __shfl_xor(sm, off, 64); }
    // the wave-level counts are only defined in lanes that traversed; take the largest
    uint32_t wv[5] = {t.wInt, t.wIntLanes, t.wLeaf, t.wLeafLanes, t.wSvc};
    for (int k = 0; k < 5; k++) {
        uint32_t v = iters ? wv[k] : 0;
        for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
        wv[k] = v;
    }
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* w = P.waveProfile + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8;
        w[0] = tStart; w[1] = wall_clock64(); w[2] = mx; w[3] = sm;
        w[4] = wv[0]; w[5] = wv[1]; w[6] = wv[2]; w[7] = (unsigned long long)wv[3] | ((unsigned long long)wv[4] << 32);
    }
}
enum { ST_RAYS = 0, ST_POPS, ST_BVCHECKS, ST_OBJVISITS, ST_TRIITERS, ST_TRICHECKS, ST_NEIGHBOURS, ST_ACTIVE };

DEV uint32_t* laneStack(uint32_t* lds, int depth) {
    uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    return lds + (size_t)wave * depth * 64 + lane;
}

DEV void storeHit(const KParams& P, int64_t i, const HitInfo& h) {
    P.hit.p0[i] = make_float4(h.hitPoint.x, h.hitPoint.y, h.hitPoint.z, h.dst);
    P.hit.p1[i] = make_float4(h.normal.x, h.normal.y, h.normal.z, __int_as_float(h.materialIndex));
    P.hit.p2[i] = make_float4(h.incomingDir.x, h.incomingDir.y, h.incomingDir.z, h.uv.x);
    P.hit.p3[i] = h.uv.y;
}
DEV HitInfo loadHit(const KParams& P, int64_t i) {
    float4 a = P.hit.p0[i], b = P.hit.p1[i], c = P.hit.p2[i];
    HitInfo h;
    h.dst = a.w; h.hitPoint = V3(a.x, a.y, a.z);
    h.normal = V3(b.x, b.y, b.z); h.materialIndex = __float_as_int(b.w);
    h.incomingDir = V3(c.x, c.y, c.z); h.uv = {c.w, P.hit.p3[i]};
    return h;
}
struct Reservoir { v3 dir, hitPoint, normal; float brdf, chosenWeight, totWeights, totCandidates; };
DEV Reservoir emptyReservoir() {
    Reservoir r;
    r.dir = r.hitPoint = r.normal = V3(0, 0, 0);
    r.brdf = r.chosenWeight = r.totWeights = r.totCandidates = 0;
    return r;
}
DEV void storeReservoir(const Recs& rc, int64_t i, const Reservoir& r) {
    if (rc.tag) rc.tag[i] = rc.tagValue;
    float4* q = rc.q + i * 4;
    q[0] = make_float4(r.dir.x, r.dir.y, r.dir.z, r.chosenWeight);
    q[1] = make_float4(r.hitPoint.x, r.hitPoint.y, r.hitPoint.z, r.totWeights);
    q[2] = make_float4(r.normal.x, r.normal.y, r.normal.z, r.totCandidates);
    q[3] = make_float4(r.brdf, 0.0f, 0.0f, 0.0f);
}
DEV Reservoir unpackReservoir(float4 a, float4 b, float4 c, float4 d) {
    Reservoir r;
    r.dir = V3(a.x, a.y, a.z); r.chosenWeight = a.w;
    r.hitPoint = V3(b.x, b.y, b.z); r.totWeights = b.w;
    r.normal = V3(c.x, c.y, c.z); r.totCandidates = c.w;
    r.brdf = d.x;
    return r;
}
DEV void storeReservoir(const Planes& pl, int64_t i, const Reservoir& r) {
    pl.p0[i] = make_float4(r.dir.x, r.dir.y, r.dir.z, r.chosenWeight);
    pl.p1[i] = make_float4(r.hitPoint.x, r.hitPoint.y, r.hitPoint.z, r.totWeights);
    pl.p2[i] = make_float4(r.normal.x, r.normal.y, r.normal.z, r.totCandidates);
    pl.p3[i] = r.brdf;
}
DEV Reservoir loadReservoir(const Planes& pl, int64_t i) {
    if (i < 0 || i >= pl.count) return emptyReservoir();
    return unpackReservoir(pl.p0[i], pl.p1[i], pl.p2[i], make_float4(pl.p3[i], 0.0f, 0.0f, 0.0f));
}
// index beyond the buffer -> zero reservoir
DEV Reservoir loadReservoir(const Recs& rc, int64_t i) {
    if (i < 0 || i >= rc.count) return emptyReservoir();
    const float4* q = rc.q + i * 4;
    return unpackReservoir(q[0], q[1], q[2], q[3]);
}

// A wave reading or writing the 64-byte reservoir records of ITS OWN 8x8 tile: one lane per record would touch 32
// cache lines per 16-byte instruction (records are 64 bytes apart), so four adjacent lanes move the four quarters of
// one record - every instruction covers two image rows of the tile as two runs of 512 contiguous bytes - and the wave
// transposes through LDS (`stage`: 64 records of 5 float4, the fifth is padding against bank conflicts).
// `mask` = lanes whose pixel takes part; x0, y0 = the tile's first pixel.
constexpr int COOP_ROW = 5;
DEV void waveFenceLds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DEV Reservoir coopLoadReservoir(const KParams& P, const Recs& rc, int x0, int y0, unsigned long long mask, float4* stage) {
    const int lane = (int)(threadIdx.x & 63), col = (lane & 31) >> 2, piece = lane & 3;
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const int row = 2 * k + (lane >> 5), owner = row * 8 + col;
        if ((mask >> owner) & 1ull) {
            const int64_t i = (int64_t)((uint32_t)P.ResX * (uint32_t)(y0 + row) + (uint32_t)(x0 + col));
            stage[owner * COOP_ROW + piece] = rc.q[i * 4 + piece];
        }
    }
    waveFenceLds();
    Reservoir r = emptyReservoir();
    if ((mask >> lane) & 1ull) {
        const float4* q = stage + lane * COOP_ROW;
        r = unpackReservoir(q[0], q[1], q[2], q[3]);
    }
    waveFenceLds();
    return r;
}
DEV void coopStoreReservoir(const KParams& P, const Recs& rc, int x0, int y0, unsigned long long mask, float4* stage, const Reservoir& r) {
    const int lane = (int)(threadIdx.x & 63), col = (lane & 31) >> 2, piece = lane & 3;
    if ((mask >> lane) & 1ull) {
        float4* q = stage + lane * COOP_ROW;
        q[0] = make_float4(r.dir.x, r.dir.y, r.dir.z, r.chosenWeight);
        q[1] = make_float4(r.hitPoint.x, r.hitPoint.y, r.hitPoint.z, r.totWeights);
        q[2] = make_float4(r.normal.x, r.normal.y, r.normal.z, r.totCandidates);
        q[3] = make_float4(r.brdf, 0.0f, 0.0f, 0.0f);
    }
    waveFenceLds();
    #pragma unroll
    for (int k = 0; k < 4; k++) {
        const int row = 2 * k + (lane >> 5), owner = row * 8 + col;
        if ((mask >> owner) & 1ull) {
            const int64_t i = (int64_t)((uint32_t)P.ResX * (uint32_t)(y0 + row) + (uint32_t)(x0 + col));
            rc.q[i * 4 + piece] = stage[owner * COOP_ROW + piece];
            if (rc.tag && piece == 0) rc.tag[i] = rc.tagValue;
        }
    }
    waveFenceLds();
}
DEV float depthAt(const KParams& P, int x, int y) {
    if (x < 0 || y < 0 || x >= P.ResX || y >= P.ResY) return 0.0f;
    return P.depth.r[(int64_t)y * P.ResX + x];
}
DEV v3 GetFocalPos(const KParams& P, v2 uv) {  // RS:441-449
    float focalPosX = (uv.x - 0.5f) * 2.0f * P.ViewSpaceDimsX;
    float focalPosY = (uv.y - 0.5f) * 2.0f * P.ViewSpaceDimsY;
    v3 focalPos = V3(focalPosX, focalPosY, 1) * P.FocalPlaneFactor;
    return mul33(P.camT, focalPos) + V3(P.camPos[0], P.camPos[1], P.camPos[2]);
}
DEV v2 GetLastUV(const KParams& P, v3 worldFocalPos) {  // RS:451-461
    v3 focalPos = worldFocalPos - V3(P.lastCamPos[0], P.lastCamPos[1], P.lastCamPos[2]);
    v3 cameraFocalPos = mul33(P.lastCamTInv, focalPos);
    v3 ndc = cameraFocalPos / P.FocalPlaneFactor;
    float uvX = (ndc.x / (2.0f * P.ViewSpaceDimsX)) + 0.5f;
    float uvY = (ndc.y / (2.0f * P.ViewSpaceDimsY)) + 0.5f;
    return {clampf(uvX, 0.0f, 1.0f), clampf(uvY, 0.0f, 1.0f)};
}

// =====================================================================================================
// InitialTrace, first half (RS:463-491,496-509): camera ray, closest hit, G-buffer outputs.
// =====================================================================================================
#ifndef RT_PRIMARY_WAVES
#define RT_PRIMARY_WAVES 4   // 101 registers, nothing spilled.  (5 = 96 registers with 5 values spilled was the better choice while the
#endif                     // kernel ran in workgroups of four waves; in single waves 4 gives 157 against 166 us, 6 gives 190.)
// The work of one workgroup of the launch: `launchBlock` = its entry of the block table.
template <bool INSTR>
DEV void primaryBlock(const KParams& P, uint32_t launchBlock, uint32_t* lds) {
    const Pixel px = pixelOfThread(P, launchBlock);
    uint32_t rays = 0;
    Trav t;
    t.dst = INF();
    if (INSTR) { t.pops = t.objVisits = t.triIters = t.iters = 0; t.triChecks = t.bvChecks = 0; }
    travResetWaveCounts<INSTR>(t);
    unsigned long long tStart = 0;
    if (INSTR && P.waveProfile) tStart = wall_clock64();
    if (px.active) {
        const uint32_t idx = (uint32_t)px.x, idy = (uint32_t)px.y;
        uint32_t state = idy * (uint32_t)P.ResX + idx + P.FrameRand;
        v2 uv = {(float)idx / (float)P.ResX, (float)idy / (float)P.ResY};
        v3 worldFocalPos = GetFocalPos(P, uv);
        v2 c = randPointInCircle(state);
        c.x *= P.DefocusStrength; c.y *= P.DefocusStrength;
        v3 jitter = mul33(P.camT, V3(c.x, c.y, 0));
        v3 rpos = V3(P.camPos[0], P.camPos[1], P.camPos[2]) + jitter * P.FocalPlaneFactor;
        v3 rdir = normalize(worldFocalPos - rpos);

        TRAVERSE<INSTR>(P, rpos, rdir, rcp3(rdir), laneStack(lds, P.stackDepth), t);
        rays = 1;
        HitInfo h = resolveHit(P, t, rpos, rdir);
        const bool didHit = h.dst != INF();
        h.incomingDir = rdir;
        rt_material material = loadMaterial(P, h.materialIndex);
        float bump = material.bump == -1 ? SampleAtlas(P, h.uv, material.bumpTexLoc, material.bumpTexDims).x : material.bump;
        h.hitPoint = h.hitPoint - h.incomingDir * bump;

        const int64_t i = (int64_t)idy * P.ResX + idx;
        storeHit(P, i, h);
        const bool badMaxBVHDepth = P.MaxBVHDepth > MAX_BVH_DEPTH;
        storeRT(P.debugOverlay, i, make_float4(t.triChecks / (float)P.DebugMaxX, t.bvChecks / (float)P.DebugMaxY,
                                               badMaxBVHDepth ? 1.0f : 0.0f, 0.0f));
        P.depth.r[i] = h.dst;
        storeRT(P.normals, i, make_float4((h.normal.x + 1) * 0.5f, (h.normal.y + 1) * 0.5f, (h.normal.z + 1) * 0.5f, 0.0f));
        v3 firstHit = didHit ? h.hitPoint : V3(0, 0, 0);
        float4 fh = make_float4(firstHit.x, firstHit.y, firstHit.z, 0.0f);
        storeRT(P.FrameCount % 2 == 0 ? P.hitA : P.hitB, i, fh);
        if (INSTR && P.records) {
            uint32_t* r = P.records + i * RECORD_DWORDS;
            r[0] = (uint32_t)t.tri; r[1] = (uint32_t)t.obj; r[2] = t.hash; r[3] = 0xffffffffu; r[4] = 0xffffffffu;
            r[5] = 0; r[6] = 0xffffffffu; r[7] = 0;
        }
    }
    writeTileCost(P, px.active && t.dst != INF(), launchBlock);
    if (INSTR && P.waveProfile && P.waveProfileKernel == 0) writeWaveProfile(P, tStart, px.active ? t.iters : 0, t);
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
// BG ("background"): the launch has only a few resident workgroups (P.bgGroups, a multiple of 8), each of which renders
// block after block of the table until the table is used up, so that the primary rays of the NEXT frame can run for the
// whole length of the frame before without taking more than a fraction of the chip's wave slots from its kernels.
// Workgroup w runs on XCD w mod 8 and takes the table entries 8k + (w mod 8), k from a counter of that XCD: the entries
// the block table has laid out for that XCD (rt_native.hip: buildBlockMap, k_order_blocks), in its order.
template <bool INSTR, bool BG = false>
__global__ __launch_bounds__(256, RT_PRIMARY_WAVES) void k_primary(const KParams P) {
    extern __shared__ uint32_t lds[];
    if (!BG) {
        primaryBlock<INSTR>(P, tableEntryOfGroup(), lds);
        return;
    }
    __shared__ uint32_t nextEntry;
    const uint32_t xcd = blockIdx.x & 7u;
    for (;;) {   // ends for every workgroup: the counter only grows, and P.bgBlocks is fixed
        if (threadIdx.x == 0) nextEntry = atomicAdd(&P.bgCounter[xcd], 1u) * 8u + xcd;
        __syncthreads();
        const uint32_t entry = nextEntry;
        __syncthreads();   // (all four waves have read it before the next round overwrites it)
        if (entry >= P.bgBlocks) break;
        primaryBlock<INSTR>(P, entry, lds);
    }
}

// =====================================================================================================
// InitialTrace, second half (RS:494 -> GenerateCandidatesDL, RS:385-434): one RIS light sample per hit pixel
// and, under VIS_REUSE, its shadow ray.
// =====================================================================================================
DEV int LightObjectReservoirSample(const KParams& P, uint32_t& state) {  // RS:293-313
    auto area = [&](int i) { return (i >= 0 && (int64_t)i < P.nLights) ? P.lights[i].areaApprox : 0.0f; };
    int firstIndex = randIntSpan(0, (int)P.EmittingObjectsNum, state);
    int chosenIndex = firstIndex;
    float totWeights = area(firstIndex);
    for (uint32_t i = 0; i < P.SceneObjectReservoirTestsNum; i++) {
        int candidateIndex = randIntSpan(0, (int)P.EmittingObjectsNum, state);
        float candidateWeight = area(candidateIndex);
        if (weightedRand(candidateWeight, totWeights, state)) chosenIndex = candidateIndex;
        totWeights += candidateWeight;
    }
    return chosenIndex;
}
DEV int TriReservoirSample(const KParams& P, uint32_t startTriIndex, uint32_t endTriIndex, uint32_t& state) {  // RS:328-348
    auto area = [&](int i) { return (i >= 0 && (int64_t)i < P.nTris) ? P.tris[i].area : 0.0f; };
    int firstIndex = randIntSpan((int)startTriIndex, (int)endTriIndex, state);
    int chosenIndex = firstIndex;
    float totWeights = area(firstIndex);
    for (uint32_t i = 0; i < P.TriReservoirTestsNum; i++) {
        int candidateIndex = randIntSpan((int)startTriIndex, (int)endTriIndex, state);
        float candidateWeight = area(candidateIndex);
        if (weightedRand(candidateWeight, totWeights, state)) chosenIndex = candidateIndex;
        totWeights += candidateWeight;
    }
    return chosenIndex;
}
DEV v2 GenerateCandidateWeight(const KParams& P, float sqrDst, float hitAngleDot, v3 incomingDir, v3 outgoingDir,
                               v3 hitNormal, float metallicity, float roughness, float brightness) {  // RS:351-360
    float weight = brightness * hitAngleDot / sqrDst;
    float brdf = 1.0f;
    if (P.kwBrdf) {
        brdf = GetBRDF(incomingDir, outgoingDir, hitNormal, metallicity, roughness);
        weight *= brdf;
    }
    return {weight, brdf};
}

// TemporalReuse for one pixel that holds a hit (RS:592-640): `thisFrame` is the pixel's reservoir of this frame.  Stores
// the pixel's TemporalFrameBuffer record; returns whether `thisFrame` (merged or not) also replaces the Candidates record.
DEV bool temporalMerge(const KParams& P, const Pixel& px, Reservoir& thisFrame) {
    const uint32_t idx = (uint32_t)px.x, idy = (uint32_t)px.y;
    uint32_t state = idy * (uint32_t)P.ResX + idx + P.FrameRand;
    const uint32_t pixelIndex = (uint32_t)P.ResX * idy + idx;
    v2 uv = {(float)idx / (float)P.ResX, (float)idy / (float)P.ResY};
    v3 worldFocalPos = GetFocalPos(P, uv);
    v2 lastUV = GetLastUV(P, worldFocalPos);
    float dx = uv.x - lastUV.x, dy = uv.y - lastUV.y;
    if (__builtin_sqrtf(dx * dx + dy * dy) > P.TemporalPrecisionThreshold) {
        storeReservoir(P.temporal, pixelIndex, thisFrame);
        return false;
    }
    const int64_t i = (int64_t)idy * P.ResX + idx;
    float4 a = P.hitA.rgba[i], b = P.hitB.rgba[i];
    float movement = length(V3(a.x, a.y, a.z) - V3(b.x, b.y, b.z));
    if (movement <= P.PixelMovementThreshold) {
        Reservoir last = loadReservoir(P.temporal, pixelIndex);
        float adjustedThis = thisFrame.totWeights * (1.0f - P.TemporalReuseWeight);
        float adjustedLast = last.totWeights * P.TemporalReuseWeight;
        float totWeights = adjustedThis + adjustedLast;
        if (weightedRand(adjustedLast, totWeights, state)) {
            thisFrame.dir = last.dir;
            thisFrame.chosenWeight = last.chosenWeight;
            thisFrame.hitPoint = last.hitPoint;
        }
        thisFrame.totWeights = avg(thisFrame.totWeights * (1.0f - P.TemporalReuseWeight), last.totWeights * P.TemporalReuseWeight);
        thisFrame.totCandidates = avg(thisFrame.totCandidates * (1.0f - P.TemporalReuseWeight), last.totCandidates * P.TemporalReuseWeight);
    }
    storeReservoir(P.temporal, pixelIndex, thisFrame);
    return true;
}
// TEMP: the launch also makes TemporalReuse's merge (RS:586-641) for its pixel, on the reservoir it has just produced
// (or, for a pixel whose depth holds a hit that its HitInfo does not, on the stored one), and stores the result once.
template <bool INSTR, bool TEMP = false, bool TOP = false>
__global__ __launch_bounds__(256, RT_MIN_WAVES) void k_candidates(const KParams P) {
    extern __shared__ uint32_t lds[];
    const float4* topLds = TOP ? loadTopTable(P, lds) : nullptr;
    const Pixel px = pixelOfThread(P);
    uint32_t rays = 0;
    Trav t;
    if (INSTR) { t.pops = t.objVisits = t.triIters = t.iters = 0; t.triChecks = t.bvChecks = 0; }
    travResetWaveCounts<INSTR>(t);
    unsigned long long tStart = 0;
    if (INSTR && P.waveProfile) tStart = wall_clock64();
    if (px.active) {
        const uint32_t idx = (uint32_t)px.x, idy = (uint32_t)px.y;
        const int64_t i = (int64_t)idy * P.ResX + idx;
        bool made = false;           // this launch produced the pixel's reservoir
        Reservoir res = emptyReservoir();
        if (P.hit.p0[i].w != INF()) {  // pixels whose primary ray missed need only this one plane
            const HitInfo hitInfo = loadHit(P, i);
            // the two draws of randPointInCircle precede this in the reference's single kernel
            uint32_t state = idy * (uint32_t)P.ResX + idx + P.FrameRand;
            NextRandom(state);
            NextRandom(state);
            const uint32_t pixelIndex = (uint32_t)P.ResX * idy + idx;

            uint32_t lightObjectIndex = (uint32_t)LightObjectReservoirSample(P, state);
            rt_light_object lightObject;
            if ((int64_t)lightObjectIndex < P.nLights) lightObject = P.lights[lightObjectIndex];
            else __builtin_memset(&lightObject, 0, sizeof lightObject);
            uint32_t startTriIndex = (uint32_t)lightObject.triStart;
            uint32_t endTriIndex = (uint32_t)(lightObject.triStart + lightObject.totTris);
            uint32_t triIndex = (uint32_t)TriReservoirSample(P, startTriIndex, endTriIndex, state);
            if (INSTR && P.records) { P.records[i * RECORD_DWORDS + 3] = lightObjectIndex; P.records[i * RECORD_DWORDS + 4] = triIndex; }
            rt_triangle tri = loadTriangle(P, (int64_t)triIndex);
            // GetRandWorldPointTri, RS:362-383
            v3 v0 = V3(loadVertex(P, tri.vertex0Index).pos), v1 = V3(loadVertex(P, tri.vertex1Index).pos),
               v2_ = V3(loadVertex(P, tri.vertex2Index).pos);
            float r1 = randNormalized(state);
            float r2 = randNormalized(state);
            if (r1 + r2 > 1.0f) { r1 = 1.0f - r1; r2 = 1.0f - r2; }
            v3 localPoint = v0 * (1.0f - r1 - r2) + v1 * r1 + v2_ * r2;
            v3 worldPoint = mul44(lightObject.localToWorldMatrix.m, localPoint, 1.0f);

            v3 dst = worldPoint - hitInfo.hitPoint;
            v3 dir = normalize(dst);
            if (dot(hitInfo.normal, dir) < 0) {
                res = emptyReservoir();
            } else {
                v3 worldNormal = normalize(mul44(lightObject.localToWorldMatrix.m, V3(tri.localNormal), 0.0f));
                float hitAngleDot = fmaxf_(dot(worldNormal, dir), 0);
                float sqrDst = dot(dst, dst);
                rt_material material = loadMaterial(P, hitInfo.materialIndex);
                float metallicity = matMetallicity(P, material, hitInfo.uv);
                float roughness = matRoughness(P, material, hitInfo.uv);
                v2 wb = GenerateCandidateWeight(P, sqrDst, hitAngleDot, hitInfo.incomingDir, dir, hitInfo.normal,
                                                metallicity, roughness, lightObject.brightness);
                if (P.kwVisReuse) {  // RS:422-430
                    TRAVERSE<INSTR, false, TOP>(P, hitInfo.hitPoint, dir, rcp3(dir), laneStack(lds, P.stackDepth), t, topLds);
                    rays = 1;
                    int shadowMat = t.obj >= 0 ? P.objs[t.obj].materialIndex : 0;
                    if (loadMaterial(P, shadowMat).brightness == 0) wb.x = 0;
                    if (INSTR && P.records) { P.records[i * RECORD_DWORDS + 5] = t.hash; P.records[i * RECORD_DWORDS + 6] = (uint32_t)t.tri; }
                }
                res.dir = dir; res.hitPoint = hitInfo.hitPoint; res.normal = hitInfo.normal;
                res.brdf = wb.y; res.chosenWeight = wb.x; res.totWeights = wb.x; res.totCandidates = 1;
            }
            made = (int64_t)pixelIndex < P.cand.count;
        }
        if (TEMP && P.FrameCount != 0 && P.depth.r[i] != INF()) {
            if (!made) res = loadReservoir(P.cand, i);
            if (temporalMerge(P, px, res)) made = true;
        }
        if (made) storeReservoir(P.cand, i, res);
    }
    if (INSTR && P.waveProfile && P.waveProfileKernel == 1) writeWaveProfile(P, tStart, rays ? t.iters : 0, t);
    if (P.stats) {
        waveAdd(P.stats + ST_RAYS, rays);
        if (INSTR) {
            waveAdd(P.stats + ST_POPS, rays ? t.pops : 0);
            waveAdd(P.stats + ST_BVCHECKS, rays ? (uint32_t)t.bvChecks : 0);
            waveAdd(P.stats + ST_OBJVISITS, rays ? t.objVisits : 0);
            waveAdd(P.stats + ST_TRIITERS, rays ? t.triIters : 0);
            waveAdd(P.stats + ST_TRICHECKS, rays ? (uint32_t)t.triChecks : 0);
        }
    }
}

// =====================================================================================================
// TemporalReuse (RS:586-641)
// =====================================================================================================
__global__ __launch_bounds__(256) void k_temporal(const KParams P) {
    __shared__ float4 stageAll[WAVES_PER_BLOCK * 64 * COOP_ROW];
    float4* stage = stageAll + (threadIdx.x >> 6) * 64 * COOP_ROW;
    const Pixel px = pixelOfThread(P);
    bool active = px.active && P.FrameCount != 0;
    if (active) active = depthAt(P, px.x, px.y) != INF();
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(active);
    if (mask != 0) {  // wave-uniform: the record moves below are made by the whole wave
        const int lane = (int)(threadIdx.x & 63);
        const int x0 = px.x - (lane & 7), y0 = px.y - (lane >> 3);
        Reservoir thisFrame = coopLoadReservoir(P, P.cand, x0, y0, mask, stage);
        bool storeCand = false;
        if (active) storeCand = temporalMerge(P, px, thisFrame);
        coopStoreReservoir(P, P.cand, x0, y0, __builtin_amdgcn_ballot_w64(storeCand), stage, thisFrame);
    }
    if (P.stats && P.statsLevel >= 2) waveAdd(P.stats + ST_ACTIVE, active ? 1u : 0u);
}

// =====================================================================================================
// SpatialReusePass (RS:514-571)
// =====================================================================================================
// Dynamic LDS: 64 * COOP_ROW float4 per wave of the workgroup (four waves, or one: see waveInBlock).
__global__ __launch_bounds__(256) void k_spatial(const KParams P) {
    extern __shared__ float4 stageAll[];
    float4* stage = stageAll + (threadIdx.x >> 6) * 64 * COOP_ROW;
    const Pixel px = pixelOfThread(P);
    bool active = px.active;
    if (active) active = depthAt(P, px.x, px.y) != INF();
    uint32_t neighbours = 0;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(active);
    const int x0 = px.x - (int)(threadIdx.x & 7), y0 = px.y - (int)((threadIdx.x & 63) >> 3);
    const Recs& src = P.ReuseBufferCycle ? P.cand : P.candB;
    const Recs& dstb = P.ReuseBufferCycle ? P.candB : P.cand;
    Reservoir sp = emptyReservoir();
    if (mask != 0) sp = coopLoadReservoir(P, src, x0, y0, mask, stage);  // the tile's own records, moved by the whole wave
    if (active) {
        const uint32_t idx = (uint32_t)px.x, idy = (uint32_t)px.y;
        uint32_t state = idy * (uint32_t)P.ResX + idx + P.FrameRand;
        const uint32_t pixelBaseIndex = (uint32_t)P.ResX * idy + idx;
        const Reservoir base = sp;
        for (uint32_t k = 0; k < P.TemporalCandidatesNum; k++) {
            int ox = rtdm::f2i((float)randIntSpan(-P.MaxOffset, P.MaxOffset, state));
            int oy = rtdm::f2i((float)randIntSpan(-P.MaxOffset, P.MaxOffset, state));
            int offX = px.x + ox, offY = px.y + oy;
            // `any(offsetID) < 0` (RS:528) is always false: negative coordinates are not rejected; a negative x
            // wraps into the previous row through the unsigned pixel index, a negative y indexes out of range.
            // The neighbour's depth and its reservoir are both asked for before the depth is tested (the reservoir
            // speculatively, from this pixel's own record when the neighbour's does not exist).  The compiler moves the
            // record's loads below the test all the same, and that order is the faster one: the pass is bound by L1
            // requests, and a neighbour without a hit then costs no record fetch (held in one batch by an asm barrier the
            // pass takes 54-56 instead of 50.5 us at 1080p).
            const bool inImage = !(offX >= P.ResX || offY >= P.ResY);
            const uint32_t pixelIndex = (uint32_t)P.ResX * (uint32_t)offY + (uint32_t)offX;
            const bool inBuffer = (int64_t)pixelIndex < src.count;  // index beyond the buffer -> zero reservoir
            const int64_t li = inImage && inBuffer ? (int64_t)pixelIndex : (int64_t)pixelBaseIndex;
            const float nDepth = depthAt(P, offX, offY);
            const float4* cq = src.q + li * 4;
            const float4 ca = cq[0], cb = cq[1], cc = cq[2], cd = cq[3];
            if (!inImage || nDepth == INF()) continue;
            const Reservoir cand = inBuffer ? unpackReservoir(ca, cb, cc, cd) : emptyReservoir();
            neighbours++;
            if (cand.dir.x == 0 || cand.dir.y == 0 || cand.dir.z == 0) continue;  // `all(dir) == 0`, RS:536
            float hitPointDiff = length(base.hitPoint - cand.hitPoint);
            // acos never exceeds 180 degrees (and a NaN fails the comparison below like any angle within the
            // threshold), so with a threshold above that the angle need not be evaluated
            float normalDiffDeg = P.SpatialNormalsAngleThreshold > 181.0f ? 0.0f : angleBetweenNormals(base.normal, cand.normal);
            if (hitPointDiff > P.SpatialHitPointDiffThreshold || normalDiffDeg > P.SpatialNormalsAngleThreshold ||
                rtdm::fabs_(sp.brdf - cand.brdf) > P.SpatialBRDFThreshold)
                continue;
            if (weightedRand(cand.chosenWeight, sp.totWeights, state)) {
                sp.dir = cand.dir; sp.chosenWeight = cand.chosenWeight; sp.hitPoint = cand.hitPoint;
                sp.normal = cand.normal; sp.brdf = cand.brdf;
            }
            sp.totWeights += cand.totWeights;
            sp.totCandidates += cand.totCandidates;
        }
    }
    if (mask != 0) coopStoreReservoir(P, dstb, x0, y0, mask, stage, sp);
    if (P.stats && P.statsLevel >= 2) { waveAdd(P.stats + ST_ACTIVE, active ? 1u : 0u); waveAdd(P.stats + ST_NEIGHBOURS, neighbours); }
}

// SpatialReusePass for small windows, with the neighbourhood staged in LDS.  The nine neighbours of a pixel lie within
// MaxOffset pixels of it, so everything the 256 pixels of a workgroup's 16x16 block can ask for is the (16 + 2 MaxOffset)^2
// region around the block: 484 records for the first pass (MaxOffset 3) against the 2,560 record fetches the block makes
// when every lane gathers for itself.  The workgroup loads the region's depths, then - in coalesced runs of whole region
// rows, four lanes per 64-byte record - the records of its hit pixels (a neighbour whose depth holds no hit is rejected
// before its reservoir is looked at, RS:531), as four planes of 16-byte quarters, and every lane then takes its
// neighbours from LDS.  The coordinates the reference does NOT reject (RS:528 is always false) - a negative x, which
// wraps into the previous image row, or a negative y - never fall into the region's in-image part and keep the global
// path of k_spatial.  Same draws, same tests, same arithmetic in the same order: bit-identical to k_spatial.
// LDS: 68 bytes per region pixel (33 KB at MaxOffset 3, 79 KB at MaxOffset 9); the region's LDS is reused for the
// transposed store of the block's own results.
DEV Reservoir spatialNeighbourGlobal(const KParams& P, const Recs& src, int offX, int offY, uint32_t pixelBaseIndex, bool inImage,
                                     float& nDepth, bool& inBuffer) {
    const uint32_t pixelIndex = (uint32_t)P.ResX * (uint32_t)offY + (uint32_t)offX;
    inBuffer = (int64_t)pixelIndex < src.count;  // index beyond the buffer -> zero reservoir
    const int64_t li = inImage && inBuffer ? (int64_t)pixelIndex : (int64_t)pixelBaseIndex;
    nDepth = depthAt(P, offX, offY);
    const float4* cq = src.q + li * 4;
    return unpackReservoir(cq[0], cq[1], cq[2], cq[3]);
}
__global__ __launch_bounds__(256) void k_spatial_lds(const KParams P) {
    extern __shared__ float4 region[];
    const int M = P.MaxOffset, RW = 16 + 2 * M, RN = RW * RW;
    float* rdepth = (float*)(region + 4 * RN);
    const uint32_t m = P.blockMap[blockIdx.x];
    if (m == 0xffffffffu) return;   // (the whole workgroup)
    const int tid = (int)threadIdx.x;
    const int rx0 = (int)(m & 0xffffu) * 16 - M, ry0 = (P.row0 & ~7) + (int)(m >> 16) * 16 - M;
    const Recs& src = P.ReuseBufferCycle ? P.cand : P.candB;
    const Recs& dstb = P.ReuseBufferCycle ? P.candB : P.cand;
    const Pixel px = pixelOfThread(P);
    const int own = (px.y - ry0) * RW + (px.x - rx0);
    const bool active = px.active && depthAt(P, px.x, px.y) != INF();
    if (!__syncthreads_or(active ? 1 : 0)) return;   // no hit pixel in the block
    // depths and records of the region in one go (a region row is one run of RW * 64 contiguous bytes); the records of
    // pixels without a hit are loaded too: nothing reads them, and waiting for the depths first costs a round trip
    for (int row = tid >> 6; row < RW; row += 4) {
        const int y = ry0 + row;
        for (int col = tid & 63; col < RW; col += 64) rdepth[row * RW + col] = depthAt(P, rx0 + col, y);
        if (y < 0 || y >= P.ResY) continue;
        const float4* rowBase = src.q + ((int64_t)y * P.ResX + rx0) * 4;
        for (int j = tid & 63; j < RW * 4; j += 64) {
            const int col = j >> 2, q = j & 3, x = rx0 + col;
            if (x >= 0 && x < P.ResX) region[q * RN + row * RW + col] = rowBase[j];
        }
    }
    __syncthreads();
    uint32_t neighbours = 0;
    Reservoir sp = emptyReservoir();
    if (active) {
        sp = unpackReservoir(region[own], region[RN + own], region[2 * RN + own], region[3 * RN + own]);
        const uint32_t idx = (uint32_t)px.x, idy = (uint32_t)px.y;
        uint32_t state = idy * (uint32_t)P.ResX + idx + P.FrameRand;
        const uint32_t pixelBaseIndex = (uint32_t)P.ResX * idy + idx;
        const Reservoir base = sp;
        for (uint32_t k = 0; k < P.TemporalCandidatesNum; k++) {
            int ox = rtdm::f2i((float)randIntSpan(-P.MaxOffset, P.MaxOffset, state));
            int oy = rtdm::f2i((float)randIntSpan(-P.MaxOffset, P.MaxOffset, state));
            int offX = px.x + ox, offY = px.y + oy;
            const bool inImage = !(offX >= P.ResX || offY >= P.ResY);
            Reservoir cand;
            if (inImage && offX >= 0 && offY >= 0 && ox >= -M && ox <= M && oy >= -M && oy <= M) {
                const int li = (offY - ry0) * RW + (offX - rx0);
                if (rdepth[li] == INF()) continue;
                cand = unpackReservoir(region[li], region[RN + li], region[2 * RN + li], region[3 * RN + li]);
            } else {   // outside the image, or one of the coordinates the reference lets through (see k_spatial)
                float nDepth;
                bool inBuffer;
                cand = spatialNeighbourGlobal(P, src, offX, offY, pixelBaseIndex, inImage, nDepth, inBuffer);
                if (!inImage || nDepth == INF()) continue;
                if (!inBuffer) cand = emptyReservoir();
            }
            neighbours++;
            if (cand.dir.x == 0 || cand.dir.y == 0 || cand.dir.z == 0) continue;  // `all(dir) == 0`, RS:536
            float hitPointDiff = length(base.hitPoint - cand.hitPoint);
            float normalDiffDeg = P.SpatialNormalsAngleThreshold > 181.0f ? 0.0f : angleBetweenNormals(base.normal, cand.normal);
            if (hitPointDiff > P.SpatialHitPointDiffThreshold || normalDiffDeg > P.SpatialNormalsAngleThreshold ||
                rtdm::fabs_(sp.brdf - cand.brdf) > P.SpatialBRDFThreshold)
                continue;
            if (weightedRand(cand.chosenWeight, sp.totWeights, state)) {
                sp.dir = cand.dir; sp.chosenWeight = cand.chosenWeight; sp.hitPoint = cand.hitPoint;
                sp.normal = cand.normal; sp.brdf = cand.brdf;
            }
            sp.totWeights += cand.totWeights;
            sp.totCandidates += cand.totCandidates;
        }
    }
    __syncthreads();   // every lane is done with the region: its LDS now stages the stores
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(active);
    if (mask != 0) {
        float4* stage = region + (threadIdx.x >> 6) * 64 * COOP_ROW;
        const int x0 = px.x - (int)(threadIdx.x & 7), y0 = px.y - (int)((threadIdx.x & 63) >> 3);
        coopStoreReservoir(P, dstb, x0, y0, mask, stage, sp);
    }
    if (P.stats && P.statsLevel >= 2) { waveAdd(P.stats + ST_ACTIVE, active ? 1u : 0u); waveAdd(P.stats + ST_NEIGHBOURS, neighbours); }
}

// TransferToOriginal (RS:575-582).  A pure copy, so the wave moves its tile's records cooperatively: four adjacent
// lanes copy the four quarters of one record, and every instruction covers two image rows of the tile as two runs of
// 512 contiguous bytes (one lane per record would touch 32 cache lines per instruction).
__global__ __launch_bounds__(256) void k_transfer(const KParams P) {
    const Pixel px = pixelOfThread(P);
    bool active = px.active;
    if (active) active = depthAt(P, px.x, px.y) != INF();
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(active);
    if (mask != 0) {
        const int lane = (int)(threadIdx.x & 63);
        const int x0 = px.x - (lane & 7), y0 = px.y - (lane >> 3);
        const int col = (lane & 31) >> 2, piece = lane & 3;
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const int row = 2 * k + (lane >> 5);
            if ((mask >> (row * 8 + col)) & 1ull) {
                const int64_t i = (int64_t)((uint32_t)P.ResX * (uint32_t)(y0 + row) + (uint32_t)(x0 + col));
                P.cand.q[i * 4 + piece] = P.candB.q[i * 4 + piece];
                if (P.cand.tag && piece == 0) P.cand.tag[i] = P.cand.tagValue;
            }
        }
    }
    if (P.stats && P.statsLevel >= 2) waveAdd(P.stats + ST_ACTIVE, active ? 1u : 0u);
}


// =====================================================================================================
// TraceRays (RS:645-815)
// =====================================================================================================
DEV v3 SampleEnvironmentMap(const KParams& P, v3 dir) {  // RS:102-117
    dir = -dir;
    float theta = rtdm::atan2_(dir.z, dir.x);
    float phi = rtdm::asin_(dir.y);
    float u = (theta / (2.0f * PI_F)) + 0.5f;
    float v = 0.5f - (phi / PI_F);
    return SampleTexture({u, v}, 0, 0, rtdm::f2u(P.envDimsX), rtdm::f2u(P.envDimsY), P.env);
}

// Four waves per SIMD: the bounce loop keeps some 110 values live around the traversal, and held to the 96 registers of
// five waves it spills 49 of them into the loop (256 us against 239 us at 1080p).
// XFER: the launch also makes TransferToOriginal's copy (RS:575-582) for its pixel - CandidatesB to Candidates where the
// depth buffer holds a hit - and takes the reservoir it shades with from that copy.  ACC: it also makes
// AccumulateFrames' running average (PostProcessor.compute:10-19) of the colour it stores.  Both are what the separate
// kernels would do to this pixel, in the same order.
template <bool INSTR, bool XFER = false, bool ACC = false, bool TOP = false>
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 4   // (3: 133 registers and no spilled value at all, 269 against 243 us)
#endif
__global__ __launch_bounds__(256, RT_TRACE_WAVES) void k_trace_rays(const KParams P) {
    extern __shared__ uint32_t lds[];
    const float4* topLds = TOP ? loadTopTable(P, lds) : nullptr;
    const Pixel px = pixelOfThread(P);
    uint32_t rays = 0, pops = 0, bvc = 0, objv = 0, trii = 0, tric = 0, iters = 0, steps = 0;
    Trav acc;
    if (INSTR) acc.wInt = acc.wIntLanes = acc.wLeaf = acc.wLeafLanes = acc.wSvc = 0;
    unsigned long long tStart = 0;
    if (INSTR && P.waveProfile) tStart = wall_clock64();
    // every lane's colour is stored once, after the wave has reconverged: lanes that take the short branches would
    // otherwise write their part of a 128-byte line long before the tracing lanes do, and the line goes out twice
    float4 colour = make_float4(0, 0, 0, 0);
    if (px.active) {
        const uint32_t idx = (uint32_t)px.x, idy = (uint32_t)px.y;
        uint32_t state = idy * (uint32_t)P.ResX + idx + P.FrameRand;
        const uint32_t pixelIndex = (uint32_t)P.ResX * idy + idx;
        const int64_t i = (int64_t)idy * P.ResX + idx;
        HitInfo hitInfo;
        hitInfo.dst = P.hit.p0[i].w;
        Reservoir cr = emptyReservoir();
        if (XFER) {
            Reservoir moved = emptyReservoir();
            const bool move = P.depth.r[i] != INF();   // k_transfer's condition; both buffers hold a record per pixel (checked at dispatch)
            if (move) { moved = loadReservoir(P.candB, i); storeReservoir(P.cand, i, moved); }
            if (hitInfo.dst != INF()) {
                hitInfo = loadHit(P, i);
                cr = move ? moved : loadReservoir(P.cand, pixelIndex);
            }
        } else if (hitInfo.dst != INF()) { hitInfo = loadHit(P, i); cr = loadReservoir(P.cand, pixelIndex); }
        if (hitInfo.dst == INF()) {  // a pixel whose primary ray missed needs only the incoming direction
            const float4 c = P.hit.p2[i];
            v3 col = SampleEnvironmentMap(P, V3(c.x, c.y, c.z));
            colour = make_float4(col.x, col.y, col.z, 0.0f);
        } else if ((cr.dir.x == 0 || cr.dir.y == 0 || cr.dir.z == 0) || dot(hitInfo.normal, cr.dir) < 0) {
            rt_material material = loadMaterial(P, hitInfo.materialIndex);
            v3 col = matCol(P, material, hitInfo.uv) * material.brightness / sqr(hitInfo.dst);
            colour = make_float4(col.x, col.y, col.z, 0.0f);
        } else {
            // Trace, RS:710-757.  ray.pos stays the primary hit point for every bounce, as in the reference.
            const v3 rpos = hitInfo.hitPoint;
            v3 rdir = cr.dir;
            const v3 rinv = rcp3(cr.dir);  // never refreshed: bounces beyond the first reuse it (RS:735,753)
            float totTravelDst = 0;
            float rc = dot(hitInfo.normal, rdir);
            v3 rayColor = V3(rc, rc, rc), directLight = V3(0, 0, 0), indirectLight = V3(0, 0, 0);
            v3 incomingDir = hitInfo.incomingDir;
            uint32_t hash = 2166136261u;
            for (uint32_t bounceCount = 0; bounceCount <= P.MaxBounceCount; bounceCount++) {
                if (hitInfo.dst == INF()) break;
                rt_material material = loadMaterial(P, hitInfo.materialIndex);
                bool invalid = (bounceCount == 1 && material.brightness <= 0) ||
                               (bounceCount == 0 && dot(hitInfo.normal, rdir) < 0);
                if (invalid) storeReservoir(P.cand, pixelIndex, emptyReservoir());
                float metallicity = matMetallicity(P, material, hitInfo.uv);
                float roughness = matRoughness(P, material, hitInfo.uv);
                if (bounceCount != 0) {
                    incomingDir = rdir;
                    (void)randNormalized(state);  // isSpecularBounce draw (RS:734); the flag itself is unused
                    v3 newDir = normalize(hitInfo.normal + randPointOnUnitSphere(state));
                    if (dot(hitInfo.normal, newDir) < 0) newDir = -newDir;
                    rdir = newDir;
                    if (P.kwWeightRecalc && bounceCount == 1 && material.brightness > 0) {  // RS:645-657
                        v2 wb = GenerateCandidateWeight(P, sqr(hitInfo.dst), fmaxf_(dot(hitInfo.normal, rdir), 0),
                                                        incomingDir, rdir, hitInfo.normal, metallicity, roughness,
                                                        material.brightness);
                        P.temporal.p0[pixelIndex].w = wb.x;  // chosenWeight
                        P.temporal.p3[pixelIndex] = wb.y;    // brdf
                    }
                }
                v3 col = matCol(P, material, hitInfo.uv);
                v3 emittedLight = col * material.brightness / sqr(hitInfo.dst);
                v3 rayColFactor = col;
                if (P.kwBrdf) rayColFactor = rayColFactor * GetBRDF(incomingDir, rdir, hitInfo.normal, metallicity, roughness);
                rayColor = rayColor * rayColFactor;
                totTravelDst += hitInfo.dst;
                if (bounceCount == 0) directLight = directLight + emittedLight * rayColor;
                else indirectLight = indirectLight + emittedLight * rayColor;
                if (bounceCount != P.MaxBounceCount) {
                    Trav t;
                    travResetWaveCounts<INSTR>(t);
                    TRAVERSE<INSTR, false, TOP>(P, rpos, rdir, rinv, laneStack(lds, P.stackDepth), t, topLds);
                    rays++;
                    steps += t.steps;
                    if (INSTR) {
                        iters += t.iters;
                        acc.wInt += t.wInt; acc.wIntLanes += t.wIntLanes; acc.wLeaf += t.wLeaf; acc.wLeafLanes += t.wLeafLanes; acc.wSvc += t.wSvc;
                        pops += t.pops; bvc += (uint32_t)t.bvChecks; objv += t.objVisits; trii += t.triIters;
                        tric += (uint32_t)t.triChecks;
                        // chain the per-ray hashes the way one running hash over all rays would evolve
                        hash = rays == 1 ? t.hash : (hash ^ t.hash) * 16777619u;
                    }
                    hitInfo = resolveHit(P, t, rpos, rdir);
                }
            }
            float avgCandidateWeight = cr.totWeights / cr.totCandidates;
            float correctionWeight = avgCandidateWeight / cr.chosenWeight;
            indirectLight = indirectLight * correctionWeight;
            v3 combined = directLight + (indirectLight / sqr(totTravelDst));
            colour = make_float4(combined.x, combined.y, combined.z, 0.0f);
            if (INSTR && P.records && rays) P.records[i * RECORD_DWORDS + 7] = hash;
        }
    }
    if (px.active) {
        const int64_t i = (int64_t)px.y * P.ResX + px.x;
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
    writeTileSteps(P, steps);
    if (INSTR && P.waveProfile && P.waveProfileKernel == 2) writeWaveProfile(P, tStart, iters, acc);
    if (P.stats) {
        waveAdd(P.stats + ST_RAYS, rays);
        if (INSTR) {
            waveAdd(P.stats + ST_ACTIVE, px.active ? 1u : 0u);
            waveAdd(P.stats + ST_POPS, pops); waveAdd(P.stats + ST_BVCHECKS, bvc); waveAdd(P.stats + ST_OBJVISITS, objv);
            waveAdd(P.stats + ST_TRIITERS, trii); waveAdd(P.stats + ST_TRICHECKS, tric);
        }
    }
}

// AccumulateFrames (PostProcessor.compute:10-19)
__global__ __launch_bounds__(256) void k_accumulate(const KParams P) {
    const Pixel px = pixelOfThread(P);
    if (!px.active) return;
    const int64_t i = (int64_t)px.y * P.ResX + px.x;
    float4 n = P.result.rgba[i], a = P.accumulated.rgba[i];
    float weight = 1.0f / (float)(P.ppFrameCount + 1);
    float4 o;
    o.x = saturate(a.x * (1 - weight) + n.x * weight);
    o.y = saturate(a.y * (1 - weight) + n.y * weight);
    o.z = saturate(a.z * (1 - weight) + n.z * weight);
    o.w = 0.0f;
    storeRT(P.accumulated, i, o);
}

// CalcTriNormals (PreCalc.compute:11-21,39-50)
__global__ __launch_bounds__(256) void k_calc_tri_normals(rt_triangle* tris, int64_t nTris, const rt_vertex* verts,
                                                          int64_t nVerts, uint32_t trianglesNum) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)trianglesNum || i >= nTris) return;
    rt_triangle tri = tris[i];
    auto pos = [&](int k) { return (k >= 0 && (int64_t)k < nVerts) ? V3(verts[k].pos) : V3(0, 0, 0); };
    v3 v0 = pos(tri.vertex0Index), v1 = pos(tri.vertex1Index), v2_ = pos(tri.vertex2Index);
    v3 n = normalize(cross(v1 - v0, v2_ - v0));
    tris[i].localNormal = {n.x, n.y, n.z};
}

// ---------------- scene preparation kernels ----------------
DEV uint32_t entryOf(const rt_bv* BVs, int64_t nBVs, int64_t c) {
    if (c < 0 || c >= nBVs) return ENTRY_LEAF;  // empty leaf (rejected earlier by host validation)
    int a = BVs[c].indexA, b = BVs[c].indexB;
    if (a >= 0) {
        if (b >= 0 && b <= ENTRY_INLINE_MAX_COUNT && a <= ENTRY_INLINE_MAX_START) return ENTRY_LEAF | ((uint32_t)b << 26) | (uint32_t)a;
        return ENTRY_LEAF_INDIRECT | (uint32_t)c;
    }
    return (uint32_t)c;
}
__global__ void k_build_wide(const rt_bv* BVs, int64_t nBVs, float4* wide) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nBVs) return;
    rt_bv bv = BVs[n];
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
    if (bv.indexA < 0) {
        int64_t cA = -(int64_t)bv.indexA, cB = bv.indexB < 0 ? -(int64_t)bv.indexB : (int64_t)bv.indexB;
        rt_bv A, B;
        __builtin_memset(&A, 0, sizeof A);
        B = A;
        if (cA < nBVs) A = BVs[cA];
        if (cB < nBVs) B = BVs[cB];
        q0 = make_float4(A.min.x, A.min.y, A.max.x, A.max.y);
        q1 = make_float4(B.min.x, B.min.y, B.max.x, B.max.y);
        q2 = make_float4(A.min.z, B.min.z, A.max.z, B.max.z);
        q3 = make_float4(__uint_as_float(entryOf(BVs, nBVs, cA)), __uint_as_float(entryOf(BVs, nBVs, cB)), 0, 0);
    }
    wide[n * 4 + 0] = q0; wide[n * 4 + 1] = q1; wide[n * 4 + 2] = q2; wide[n * 4 + 3] = q3;
}
// For the host-side validation of a scene: the two index words of every node (a quarter of the node array), and
// whether any triangle names a vertex outside the vertex buffer (flag[0] set to 1).
__global__ void k_extract_links(const rt_bv* BVs, int64_t nBVs, int2* links) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nBVs) return;
    links[n] = make_int2(BVs[n].indexA, BVs[n].indexB);
}
// A digest of links[0, n) that depends on every word and its position: two sums of differently mixed (index, link)
// pairs, added into out[0] and out[1] (zeroed by the caller).  A reload uploads the same mesh BVHs again; equal
// digests of the part below the scene BVH let the host keep what it found out about them (prepareScene).
DEV unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}
DEV unsigned long long waveSum64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, off, 64), hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
__global__ __launch_bounds__(256) void k_hash_links(const int2* links, int64_t n, unsigned long long* out) {
    unsigned long long h0 = 0, h1 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int2 l = links[i];
        const unsigned long long w = ((unsigned long long)(uint32_t)l.x << 32) | (uint32_t)l.y;
        h0 += mix64(w ^ mix64((unsigned long long)i));
        h1 += mix64((w + 0x9e3779b97f4a7c15ull) ^ mix64((unsigned long long)i * 0xd6e8feb86659fd93ull + 1));
    }
    h0 = waveSum64(h0);
    h1 = waveSum64(h1);
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, h0); atomicAdd(out + 1, h1); }
}
__global__ void k_check_tri_indices(const rt_triangle* tris, int64_t nTris, int64_t nVerts, uint32_t* flag) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTris) return;
    const rt_triangle t = tris[i];
    if (t.vertex0Index < 0 || t.vertex1Index < 0 || t.vertex2Index < 0 || t.vertex0Index >= nVerts || t.vertex1Index >= nVerts || t.vertex2Index >= nVerts)
        flag[0] = 1u;
}
__global__ void k_obj_root_entries(const rt_bv* BVs, int64_t nBVs, const rt_scene_object* objs, int64_t nObjs, uint32_t* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nObjs) return;
    out[i] = entryOf(BVs, nBVs, (int64_t)(uint32_t)objs[i].bvStartIndex);
}
// The table of top nodes (ENTRY_TOP): slot s is the wide record of node plan.node[s], with the entries of those children
// that have a slot of their own replaced by that slot; and the objects' root entries with the same replacement.
__global__ void k_build_top(const float4* wide, TopPlan plan, float4* top) {
    const int s = (int)threadIdx.x;
    if (s >= plan.n) return;
    const float4* rec = wide + (size_t)plan.node[s] * 4;
    float4 q3 = rec[3];
    if (plan.slotA[s] >= 0) q3.x = __uint_as_float(ENTRY_TOP | (uint32_t)plan.slotA[s]);
    if (plan.slotB[s] >= 0) q3.y = __uint_as_float(ENTRY_TOP | (uint32_t)plan.slotB[s]);
    top[s * 4 + 0] = rec[0]; top[s * 4 + 1] = rec[1]; top[s * 4 + 2] = rec[2]; top[s * 4 + 3] = q3;
}
__global__ void k_obj_root_entries_top(const uint32_t* objRoot, const rt_scene_object* objs, int64_t nObjs, TopPlan plan, uint32_t* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nObjs) return;
    uint32_t e = objRoot[i];
    if (e < ENTRY_LEAF_INDIRECT)   // an internal root: its BV index may have a slot
        for (int s = 0; s < plan.n; s++) if (plan.node[s] == e) { e = ENTRY_TOP | (uint32_t)s; break; }
    out[i] = e;
}
__global__ void k_bake_tris(const rt_triangle* tris, int64_t nTris, const rt_vertex* verts, int64_t nVerts, float4* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTris) return;
    rt_triangle tri = tris[i];
    auto pos = [&](int k) { return (k >= 0 && (int64_t)k < nVerts) ? V3(verts[k].pos) : V3(0, 0, 0); };
    v3 v0 = pos(tri.vertex0Index), v1 = pos(tri.vertex1Index), v2_ = pos(tri.vertex2Index);
    v3 eAB = v1 - v0, eAC = v2_ - v0;
    v3 N = cross(eAB, eAC);
    out[i * 4 + 0] = make_float4(v0.x, v0.y, v0.z, N.x);
    out[i * 4 + 1] = make_float4(eAB.x, eAB.y, eAB.z, N.y);
    out[i * 4 + 2] = make_float4(eAC.x, eAC.y, eAC.z, N.z);
    out[i * 4 + 3] = make_float4(tri.localNormal.x, tri.localNormal.y, tri.localNormal.z, 0.0f);
}

// 52-byte ABI records <-> the internal layouts (see rt_device.h).  kind 0 = CandidateReservoir: dir 0-2, hitPoint 3-5,
// normal 6-8, brdf 9, chosenWeight 10, totWeights 11, totCandidates 12; kind 1 = HitInfo: dst 0, hitPoint 1-3, uv 4-5,
// normal 6-8, incomingDir 9-11, materialIndex 12.
__global__ void k_aos_to_planes(const float* aos, Planes pl, int kind) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pl.count) return;
    const float* r = aos + i * 13;
    if (kind == 0) {
        pl.p0[i] = make_float4(r[0], r[1], r[2], r[10]);
        pl.p1[i] = make_float4(r[3], r[4], r[5], r[11]);
        pl.p2[i] = make_float4(r[6], r[7], r[8], r[12]);
        pl.p3[i] = r[9];
    } else {
        pl.p0[i] = make_float4(r[1], r[2], r[3], r[0]);
        pl.p1[i] = make_float4(r[6], r[7], r[8], r[12]);
        pl.p2[i] = make_float4(r[9], r[10], r[11], r[4]);
        pl.p3[i] = r[5];
    }
}
// The two *_to_aos kernels convert records [first, first + n) into `aos` (n records).
__global__ void k_planes_to_aos(Planes pl, float* aos, int kind, int64_t first, int64_t n) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t i = first + j;
    float* r = aos + j * 13;
    float4 a = pl.p0[i], b = pl.p1[i], c = pl.p2[i];
    float d = pl.p3[i];
    if (kind == 0) {
        r[0] = a.x; r[1] = a.y; r[2] = a.z; r[10] = a.w;
        r[3] = b.x; r[4] = b.y; r[5] = b.z; r[11] = b.w;
        r[6] = c.x; r[7] = c.y; r[8] = c.z; r[12] = c.w;
        r[9] = d;
    } else {
        r[1] = a.x; r[2] = a.y; r[3] = a.z; r[0] = a.w;
        r[6] = b.x; r[7] = b.y; r[8] = b.z; r[12] = b.w;
        r[9] = c.x; r[10] = c.y; r[11] = c.z; r[4] = c.w;
        r[5] = d;
    }
}
__global__ void k_aos_to_recs(const float* aos, Recs rc) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rc.count) return;
    const float* r = aos + i * 13;
    float4* q = rc.q + i * 4;
    q[0] = make_float4(r[0], r[1], r[2], r[10]);
    q[1] = make_float4(r[3], r[4], r[5], r[11]);
    q[2] = make_float4(r[6], r[7], r[8], r[12]);
    q[3] = make_float4(r[9], 0.0f, 0.0f, 0.0f);
}
// `others`: the buffer's other copies and their tags; a pixel is taken from the copy that holds its newest record (the
// largest tag; the current copy on a tie).
DEV int newestCopy(const Recs& rc, const OtherCopies& others, int64_t i, uint32_t* tagOut) {
    int best = -1;
    uint32_t t = rc.tag ? rc.tag[i] : 0u;
    if (rc.tag)
        for (int k = 0; k < others.n; k++) {
            const uint32_t tk = others.tag[k][i];
            if (tk > t) { t = tk; best = k; }
        }
    *tagOut = t;
    return best;
}
__global__ void k_recs_to_aos(Recs rc, OtherCopies others, float* aos, int64_t first, int64_t n) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float* r = aos + j * 13;
    uint32_t t;
    const int from = newestCopy(rc, others, first + j, &t);
    const float4* q = (from < 0 ? rc.q : others.q[from]) + (first + j) * 4;
    float4 a = q[0], b = q[1], c = q[2], d = q[3];
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[10] = a.w;
    r[3] = b.x; r[4] = b.y; r[5] = b.z; r[11] = b.w;
    r[6] = c.x; r[7] = c.y; r[8] = c.z; r[12] = c.w;
    r[9] = d.x;
}
// Makes the copy `rc` of a buffer with several copies complete for tagged groups [first, first + n): a group whose
// newest contents are in another copy gets them, and the tag.  `group` = float4s per tag: 4 for a reservoir record, the
// candidates of one pixel ray for the RIS project's Candidates.  (Run with no frame in flight.)
__global__ void k_recs_merge(Recs rc, OtherCopies others, int64_t first, int64_t n, int group) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n * group) return;
    const int64_t i = first + j / group;
    const int piece = (int)(j % group);
    uint32_t t;
    const int from = newestCopy(rc, others, i, &t);
    if (from < 0) return;
    rc.q[i * group + piece] = others.q[from][i * group + piece];
}
// Second step: the tags (a kernel of its own: the threads of a large group are not all in one wave).
__global__ void k_recs_merge_tags(Recs rc, OtherCopies others, int64_t first, int64_t n) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t t;
    if (newestCopy(rc, others, first + j, &t) >= 0) rc.tag[first + j] = t;
}
// Rebuilds the workgroup -> block table with the heaviest regions first (longest-processing-time-first: the launches
// that follow end with light regions instead of with a few heavy waves on an otherwise idle chip).
// Regions (RS x RS blocks; 64x64 pixels at RS = 4) are ranked by their hit pixels in steps of 64 (a counting sort; the
// order inside a step is whatever the atomics give: any permutation renders the same image) and dealt to the 8 XCDs in
// that order; every XCD still walks whole regions.  One workgroup per table: workgroup b writes `out.t[b]`, which has
// ceil(regions / 8) * 8 * RS * RS entries, from that table's counts.
#ifndef RT_REGION
#define RT_REGION 4   // blocks per side of a region, where nothing else is asked for
#endif
constexpr int ORDER_MAX_REGIONS = 8192;   // (7 bytes of LDS per region in k_order_blocks: 57 KB)
struct OrderTables {
    struct { uint32_t* dyn; unsigned nDyn; int rs; const uint32_t* costW; int byMax; } t[3];
};
// Per table - byMax = 0: costW holds hit pixels per wave, a region's rank is their sum in steps of 64; byMax = 1: costW holds the
// steps each wave needed for its secondary rays, a region's rank is its longest wave in steps of 8 (the waves with the
// longest rays must start first: a wave cannot end before its longest ray does).
// Dynamic LDS: orderLdsBytes(cap), cap = the most regions any table of the launch has (a workgroup that asks for the
// full 57 KB waits for a CU with that much free beside the frame's kernels: 36 us per launch instead of 14).
constexpr size_t orderLdsBytes(unsigned cap) { return (size_t)((cap + 3u) & ~3u) * 7; }
constexpr int ORDER_THREADS = 1024;   // (256 or 64 threads are no sooner done beside the frame's kernels: 71 and 90 us against 62)
__global__ __launch_bounds__(ORDER_THREADS) void k_order_blocks(int BX, int BY, OrderTables out, unsigned cap) {
    const uint32_t* const costW = out.t[blockIdx.x].costW;
    const int byMax = out.t[blockIdx.x].byMax;
    constexpr uint32_t NB = 64;   // ranks (32 cost 1 % of the frame; 128 and 250 gained nothing)
    __shared__ uint32_t hist[NB + 2];
    extern __shared__ uint32_t orderLds[];
    cap = (cap + 3u) & ~3u;
    uint32_t* const regionCost = orderLds;
    uint16_t* const order = (uint16_t*)(orderLds + cap);
    uint8_t* const bucketOf = (uint8_t*)(order + cap);
    uint32_t* const dyn = out.t[blockIdx.x].dyn;
    const unsigned nDyn = out.t[blockIdx.x].nDyn;
    const int RS = out.t[blockIdx.x].rs;
    const int RX = (BX + RS - 1) / RS, RY = (BY + RS - 1) / RS, R = RX * RY;
    const int t = (int)threadIdx.x;
    if (t < (int)NB + 2) hist[t] = 0;
    for (int r = t; r < R; r += ORDER_THREADS) regionCost[r] = 0;
    __syncthreads();
    for (int b = t; b < BX * BY; b += ORDER_THREADS) {  // one 16-byte load per block: its four waves' hit counts
        const uint4 w = ((const uint4*)costW)[b];
        uint32_t* rc = &regionCost[(b / BX / RS) * RX + (b % BX) / RS];
        if (byMax) atomicMax(rc, max(max(w.x, w.y), max(w.z, w.w)));
        else atomicAdd(rc, w.x + w.y + w.z + w.w);
    }
    __syncthreads();
    for (int r = t; r < R; r += ORDER_THREADS) {
        const uint32_t c = byMax ? regionCost[r] * 8 : regionCost[r] * 16 / (RS * RS);  // hit counts: as if the region had 16 blocks
        const uint32_t q = c * NB / 4096, step = q > NB ? NB : q;
        bucketOf[r] = (uint8_t)(NB - step);  // bucket 0 = heaviest
        atomicAdd(&hist[NB - step], 1u);
    }
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t b = 0; b <= NB; b++) { const uint32_t n = hist[b]; hist[b] = run; run += n; }
    }
    __syncthreads();
    for (int r = t; r < R; r += ORDER_THREADS) order[atomicAdd(&hist[bucketOf[r]], 1u)] = (uint16_t)r;
    __syncthreads();
    for (unsigned m = (unsigned)t; m < nDyn; m += ORDER_THREADS) {
        const unsigned xcd = m % 8, j = m / 8, ord = (j / (RS * RS)) * 8 + xcd, k = j % (RS * RS);
        uint32_t entry = 0xffffffffu;
        if ((int)ord < R) {
            const uint32_t r = order[ord];
            const int x = (int)(r % RX) * RS + (int)(k % RS), y = (int)(r / RX) * RS + (int)(k / RS);
            if (x < BX && y < BY) entry = (uint32_t)y << 16 | (uint32_t)x;
        }
        dyn[m] = entry;
    }
}

// Test aid: occupies a stream for about `ticks` ticks of the 100 MHz wall clock (one wave; gives up after 4 ms whatever
// is asked for, so it always ends).
__global__ void k_delay(unsigned long long ticks) {
    if (ticks > 400000ull) ticks = 400000ull;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
__global__ void k_rgba8_to_f32(const uint8_t* src, float4* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[i] = make_float4(src[i * 4] / 255.0f, src[i * 4 + 1] / 255.0f, src[i * 4 + 2] / 255.0f, src[i * 4 + 3] / 255.0f);
}

}  // namespace rtn
