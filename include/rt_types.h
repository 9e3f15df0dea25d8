/* rt_types.h — buffer ABI of the ray-tracing hot path.
 *
 * Plain-old-data layouts of every structured buffer that crosses the host/device boundary.  They are
 * byte-for-byte the layouts the reference's C# host uploads (Marshal.SizeOf strides,
 * BVH_ReSTIR_DL/Assets/Scripts/ComputeHelper.cs:228) and its HLSL kernels declare
 * (BVH_ReSTIR_DL/Assets/Scripts/ComputeShaders/RendererDataTypes.hlsl:3-118,
 *  BVH_RIS_DL/Assets/Scripts/ComputeShaders/RendererDataTypes.hlsl:3-86).
 *
 * Shared by the CPU oracle, the HIP kernels, the host-side scene builder and the C-ABI; contains no code.
 * All fields are 4-byte aligned, no padding; matrices are column-major (c0..c3), as Unity's float4x4.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

typedef struct { float x, y; } rt_float2;
typedef struct { float x, y, z; } rt_float3;
typedef struct { float x, y, z, w; } rt_float4;
typedef struct { int32_t x, y; } rt_int2;
/* column-major 4x4: element (row r, column c) is m[c*4 + r] */
typedef struct { float m[16]; } rt_float4x4;

/* ---------------- ReSTIR project (BVH_ReSTIR_DL) ---------------- */

/* RendererDataTypes.hlsl:16-22 / Data types/Render/RenderBV.cs:3-9.
 * indexA >= 0: leaf, indexA = first component, indexB = component count.
 * indexA <  0: internal, children are nodes -indexA and -indexB. */
typedef struct {
    rt_float3 min;
    rt_float3 max;
    int32_t indexA;
    int32_t indexB;
} rt_bv;
RT_STATIC_ASSERT(sizeof(rt_bv) == 32, "BV is 32 bytes");

/* RendererDataTypes.hlsl:3-10 / RenderTriangle.cs */
typedef struct {
    int32_t vertex0Index;
    int32_t vertex1Index;
    int32_t vertex2Index;
    rt_float3 localNormal; /* object-space unit normal, written by CalcTriNormals */
    float area;
} rt_triangle;
RT_STATIC_ASSERT(sizeof(rt_triangle) == 28, "Triangle is 28 bytes");

/* RendererDataTypes.hlsl:11-15 / Vertex.cs */
typedef struct {
    rt_float3 pos;
    rt_float2 uv;
} rt_vertex;
RT_STATIC_ASSERT(sizeof(rt_vertex) == 20, "Vertex2 is 20 bytes");

/* RendererDataTypes.hlsl:23-33 / SceneObjectData.cs */
typedef struct {
    rt_float4x4 worldToLocalMatrix;
    rt_float4x4 localToWorldMatrix;
    int32_t materialIndex;
    int32_t bvStartIndex;
    int32_t maxDepthBVH;
    float areaApprox;
    rt_float3 min;
    rt_float3 max;
} rt_scene_object;
RT_STATIC_ASSERT(sizeof(rt_scene_object) == 168, "SceneObject is 168 bytes");

/* RendererDataTypes.hlsl:34-41 / LightObject.cs */
typedef struct {
    rt_float4x4 localToWorldMatrix;
    float areaApprox;
    float brightness;
    int32_t triStart;
    int32_t totTris;
} rt_light_object;
RT_STATIC_ASSERT(sizeof(rt_light_object) == 80, "LightObject is 80 bytes");

/* RendererDataTypes.hlsl:42-69 / Material2.cs.  A scalar of -1 (col: all three -1) means "sample the atlas";
 * normalsTexLoc == (-1,-1) means "no normal map". */
typedef struct {
    float brightness;
    rt_float3 col;
    rt_int2 colTexLoc;
    rt_int2 colTexDims;
    float bump;
    rt_int2 bumpTexLoc;
    rt_int2 bumpTexDims;
    float roughness;
    rt_int2 roughnessTexLoc;
    rt_int2 roughnessTexDims;
    float metallicity;
    rt_int2 metallicityTexLoc;
    rt_int2 metallicityTexDims;
    rt_int2 normalsTexLoc;
    rt_int2 normalsTexDims;
} rt_material;
RT_STATIC_ASSERT(sizeof(rt_material) == 108, "Material2 is 108 bytes");

/* RendererDataTypes.hlsl:79-87 / HitInfo.cs */
typedef struct {
    float dst; /* +INF: the primary ray missed */
    rt_float3 hitPoint;
    rt_float2 uv;
    rt_float3 normal;
    rt_float3 incomingDir;
    int32_t materialIndex;
} rt_hit_info;
RT_STATIC_ASSERT(sizeof(rt_hit_info) == 52, "HitInfo is 52 bytes");

/* RendererDataTypes.hlsl:109-118 / CandidateReservoir.cs */
typedef struct {
    rt_float3 dir;
    rt_float3 hitPoint;
    rt_float3 normal;
    float brdf;
    float chosenWeight;
    float totWeights;
    float totCandidates;
} rt_reservoir;
RT_STATIC_ASSERT(sizeof(rt_reservoir) == 52, "CandidateReservoir is 52 bytes");

/* ---------------- RIS project (BVH_RIS_DL) ---------------- */

/* BVH_RIS_DL RendererDataTypes.hlsl:3-14 */
typedef struct {
    rt_float3 vA, vB, vC;
    rt_float2 uvA, uvB, uvC;
    rt_float3 worldNormal; /* written by the RIS CalcTriNormals */
    int32_t parentIndex;
} rt_ris_tri;
RT_STATIC_ASSERT(sizeof(rt_ris_tri) == 76, "RIS Tri is 76 bytes");

/* BVH_RIS_DL RendererDataTypes.hlsl:41-47 */
typedef struct {
    rt_float3 color;
    rt_float3 specularColor;
    float brightness;
    float smoothness;
} rt_ris_material;
RT_STATIC_ASSERT(sizeof(rt_ris_material) == 32, "RIS Material2 is 32 bytes");

/* BVH_RIS_DL RendererDataTypes.hlsl:56-63 */
typedef struct {
    float dst;
    rt_float3 hitPoint;
    rt_float2 uv;
    rt_float3 normal;
    int32_t materialIndex;
} rt_ris_hit_info;
RT_STATIC_ASSERT(sizeof(rt_ris_hit_info) == 40, "RIS HitInfo is 40 bytes");

/* The RIS project's BoundingVolume, SceneObject and LightObject are byte-identical to rt_bv,
 * rt_scene_object and rt_light_object; its Candidates buffer is an array of rt_float4 (dir, weight). */

#endif /* RT_TYPES_H */
