// orc_math.h — fp32 vector helpers and the PCG-style RNG shared by the CPU oracle's translation units.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  These define the oracle's arithmetic conventions:
//   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;  mul(M,v) = per row, products summed left to right;
//   normalize(v) = v * (1 / sqrt(dot(v,v)));  min/max ignore a NaN operand;
//   out-of-range structured-buffer reads return an all-zero element.
#pragma once

#include <cstdint>
#include <cstring>

#include "../include/rt_detmath.h"
#include "../include/rt_types.h"

namespace orcm {

using rtdm::f2i;
using rtdm::f2u;

const float INF = __builtin_inff();
const float PI = 3.1415926f;  // Constants.hlsl:3
const unsigned MAX_BVH_DEPTH = 32;  // Constants.hlsl:14

struct f2 { float x, y; };
struct f3 { float x, y, z; };

inline f3 F3(float x, float y, float z) { return {x, y, z}; }
inline f3 F3(rt_float3 v) { return {v.x, v.y, v.z}; }
inline rt_float3 R3(f3 v) { return {v.x, v.y, v.z}; }
inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
inline f3 operator+(f3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline f3 operator-(f3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
inline f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
inline f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
inline f2 operator*(f2 a, float s) { return {a.x * s, a.y * s}; }
inline float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
inline f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(f3 a) { return rtdm::sqrt_(dot(a, a)); }
inline float length(f2 a) { return rtdm::sqrt_(dot(a, a)); }
inline f3 normalize(f3 a) { float inv = 1.0f / rtdm::sqrt_(dot(a, a)); return a * inv; }
inline float fminf_(float a, float b) { return rtdm::fmin_(a, b); }
inline float fmaxf_(float a, float b) { return rtdm::fmax_(a, b); }
inline f3 min3(f3 a, f3 b) { return {fminf_(a.x, b.x), fminf_(a.y, b.y), fminf_(a.z, b.z)}; }
inline f3 max3(f3 a, f3 b) { return {fmaxf_(a.x, b.x), fmaxf_(a.y, b.y), fmaxf_(a.z, b.z)}; }
inline float saturate(float x) { return fminf_(fmaxf_(x, 0.0f), 1.0f); }
inline float clampf(float x, float lo, float hi) { return fminf_(fmaxf_(x, lo), hi); }
inline float sqr(float a) { return a * a; }                   // RendererResources.hlsl:36-39
inline float avg(float a, float b) { return .5f * (a + b); }  // :41-44
inline float dot2(f3 a) { return dot(a, a); }                 // :54-57

// mul(float4x4, float4(v, w)).xyz with column-major storage: element (r,c) = m[c*4+r].
inline f3 mul44(const rt_float4x4& M, f3 v, float w) {
    const float* m = M.m;
    f3 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * w;
    return r;
}
// mul(float3x3, float3) with row-major m[r*3+c].
inline f3 mul33(const float* m, f3 v) {
    return {(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z,
            (m[6] * v.x + m[7] * v.y) + m[8] * v.z};
}

// Robust structured-buffer read: out of range -> all-zero element.
template <class T>
inline T ld(const T* p, int64_t n, int64_t i) {
    if (p && i >= 0 && i < n) return p[i];
    T z;
    std::memset(&z, 0, sizeof(T));
    return z;
}

// ---------------- RNG: RendererResources.hlsl:63-113 ----------------
inline uint32_t NextRandom(uint32_t& state) {
    state = state * 747796405u + 2891336453u;
    uint32_t result = ((state >> ((state >> 28) + 4)) ^ state) * 277803737u;
    result = (result >> 22) ^ result;
    return result;
}
// `u / 4294967295.0`: the literal rounds to 2^32 in fp32, and uint->float rounds to nearest even, so the
// result can be exactly 1.0 for u >= 0xFFFFFF80.
inline float randNormalized(uint32_t& state) { return (float)NextRandom(state) / 4294967296.0f; }
inline int randIntSpan(int a, int b, uint32_t& state) {
    float randNorm = randNormalized(state);
    int diff = b - a;
    int offset = f2i((float)diff * randNorm);
    return a + offset;
}
inline bool weightedRand(float a, float b, uint32_t& state) {
    float randNorm = randNormalized(state);
    float relRand = randNorm * b;
    return relRand < a;
}
inline float randValueNormalDistribution(uint32_t& state) {
    float theta = 2 * PI * randNormalized(state);
    float rho = rtdm::sqrt_(-2 * rtdm::log_(randNormalized(state)));
    return rho * rtdm::cos_(theta);
}
inline f3 randPointOnUnitSphere(uint32_t& state) {
    float x = randValueNormalDistribution(state);
    float y = randValueNormalDistribution(state);
    float z = randValueNormalDistribution(state);
    return normalize(F3(x, y, z));
}
inline f2 randPointInCircle(uint32_t& state) {
    float angle = randNormalized(state) * 2 * PI;
    f2 pointOnCircle = {rtdm::cos_(angle), rtdm::sin_(angle)};
    return pointOnCircle * rtdm::sqrt_(randNormalized(state));
}
// RendererResources.hlsl:115-123
inline float angleBetweenNormals(f3 a, f3 b) {
    float dotProduct = clampf(dot(a, b), -1.0f, 1.0f);
    float angleRadians = rtdm::acos_(dotProduct);
    return angleRadians * 57.29577951308232f;  // degrees()
}


}  // namespace orcm
