/* rt_oracle_ris.h — C interface of the CPU oracle for the RIS project's kernels (BVH_RIS_DL).
 *
 * TEST INFRASTRUCTURE ONLY, exactly like rt_oracle.h: used by tests/ and by __graft_entry__.smoke() as the
 * checker; nothing in the shipped library includes, links or calls it.
 *
 * Parity status: PARITY UNPINNED against the reference's own execution (the reference holds no golden vectors for
 * these kernels and its HLSL cannot run outside Unity).  Paths cited below are relative to
 * BVH_RIS_DL/Assets/Scripts/ComputeShaders/.
 */
#ifndef RT_ORACLE_RIS_H
#define RT_ORACLE_RIS_H

#include "rt_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Shader constants of RIS_BVH_RayTracer.compute:6-44.  CameraTransform is the upper-left 3x3 of the 4x4 the
 * host sets (Main.cs:206-213), row-major m[r*3+c]. */
typedef struct {
    int32_t TrisNum, BVsNum, MaxBVHDepth, SceneBVHStartIndex, SceneObjectsNum;
    float ScatterProbability;
    int32_t Resolution[2];
    int32_t MaxBounceCount, RaysPerPixel;
    float ViewSpaceDims[2];
    float DefocusStrength, FocalPlaneFactor;
    float WorldSpaceCameraPos[3];
    float CameraTransform[9];
    int32_t FrameRand, FrameCount;
    int32_t DebugDataMaxValues[2];
    int32_t SceneObjectCandidatesNum, TriCandidatesNum, TotCandidatesNum, EmittingObjectsNum;
    float TotArea;
    int32_t SceneObjectReservoirTestsNum, TriReservoirTestsNum, CandidateReservoirTestsNum;
    int32_t row0, row1; /* row band, as in orc_params */
    int32_t unorm8Targets; /* as in orc_params: Result, DebugOverlay, RayHitPoints hold 8-bit UNORM texels (Main.cs:324-364) */
} orc_ris_params;

typedef struct {
    const rt_bv* BVs; int64_t nBVs;
    const rt_scene_object* SceneObjects; int64_t nSceneObjects;
    const rt_light_object* LightObjects; int64_t nLightObjects;
    const rt_ris_material* Materials; int64_t nMaterials;
    rt_ris_tri* Tris; int64_t nTris; /* worldNormal written by orc_ris_calc_tri_normals */
} orc_ris_scene;

typedef struct {
    rt_ris_hit_info* HitInfos; int64_t nHitInfos;
    rt_float4* Candidates; int64_t nCandidates;
    rt_float4* Result;
    rt_float4* DebugOverlay;
    rt_float4* RayHitPoints;
    orc_trace_record* records; /* may be NULL: triIndex/objIndex/visitHash of the primary ray, bounceVisitHash */
    orc_counters* counters;    /* may be NULL; accumulated into */
} orc_ris_frame;

int orc_ris_calc_tri_normals(orc_ris_scene* scene, uint32_t trisNum, int threads);                    /* PreCalc.compute:30-41 */
int orc_ris_generate_candidates(const orc_ris_scene*, const orc_ris_params*, orc_ris_frame*, int threads); /* RIS_BVH_RayTracer.compute:384-412 */
int orc_ris_trace_rays(const orc_ris_scene*, const orc_ris_params*, orc_ris_frame*, int threads);          /* :503-559 */

/* Known-answer helpers. */
int32_t orc_ris_weighted_rand(float a, float b, uint32_t* state);                                /* RendererResources.hlsl:85-92 */
float orc_ris_ray_bv_intersect(const float pos[3], const float dir[3], const rt_bv* bv);         /* RIS_BVH_RayTracer.compute:111-124 */
int32_t orc_ris_ray_triangle_intersect(const float pos[3], const float dir[3], const rt_ris_tri* tri, float* dst,
                                       float uv[2]);                                             /* :86-109 */

#ifdef __cplusplus
}
#endif
#endif
