/* rt_oracle.h — C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a CPU restatement of the reference's HLSL kernels, used by tests/,
 * by __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker and the CPU baseline.  Nothing
 * in the shipped library includes, links or calls it.
 *
 * Parity status: the reference holds no golden vectors or tests for this path, and its HLSL cannot be
 * compiled or run outside Unity, so parity against the reference's own execution is UNPINNED.  The oracle is
 * pinned only where the reference's files allow it: the PCG hash and the transform matrices in the committed
 * 80kScene JSON dump (tests/test_oracle_kat.py), and the BVH topology statistics of that dump.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Shader constants of ReSTIR_BVH_RayTracer.compute:14-66,91-92 plus the three multi_compile keywords (:7-9).
 * float3x3 constants are the upper-left 3x3 of the 4x4 the host sets, stored row-major: m[r*3+c].
 * TextureAtlasDims / EnvironmentMapTexDims hold the RAW 32-bit words the host wrote (the reference sets these
 * float2 constants with SetInts, Main.cs:260-267, so the kernels see int bit patterns as floats). */
typedef struct {
    uint32_t TrianglesNum, BVsNum, MaxBVHDepth, SceneBVHStartIndex, SceneObjectsNum;
    float ScatterProbability;
    uint32_t MaxBounceCount;
    int32_t Resolution[2];
    float ViewSpaceDims[2];
    float DefocusStrength, FocalPlaneFactor;
    float WorldSpaceCameraPos[3], LastWorldSpaceCameraPos[3];
    float CameraTransform[9], LastCameraTransformInverse[9];
    uint32_t FrameRand, FrameCount;
    int32_t DebugDataMaxValues[2];
    uint32_t SceneObjectReservoirTestsNum, TriReservoirTestsNum;
    int32_t ReuseBufferCycle, MaxOffset;
    uint32_t TemporalCandidatesNum;
    float SpatialHitPointDiffThreshold, SpatialNormalsAngleThreshold, SpatialBRDFThreshold;
    float TemporalReuseWeight, PixelMovementThreshold, TemporalPrecisionThreshold;
    uint32_t EmittingObjectsNum;
    uint32_t TextureAtlasDimsRaw[2], EnvironmentMapTexDimsRaw[2];
    int32_t kw_VIS_REUSE, kw_WEIGHT_RECALC, kw_BRDF;
    /* Image rows [row0, row1) to execute (a row band); row1 <= 0 means the whole image.  Pixel coordinates,
     * RNG seeds and buffer indices stay global, so a band computes exactly what the full frame would there. */
    int32_t row0, row1;
    /* Non-zero: the 3- and 4-channel render targets (Result, AccumulatedResult, DebugOverlay, NormalsBuffer,
     * RayHitPointsA/B) hold 8-bit UNORM texels as the reference's `new RenderTexture(w, h, 24)` does
     * (TextureHelper.cs:144-155): every store clamps to [0, 1] and rounds to a multiple of 1/255.  The one place this
     * feeds back into the computation is TemporalReuse's hit-point movement test (ReSTIR.compute:613). */
    int32_t unorm8Targets;
} orc_params;

typedef struct {
    const rt_bv* BVs; int64_t nBVs;
    const rt_scene_object* SceneObjects; int64_t nSceneObjects;
    const rt_light_object* LightObjects; int64_t nLightObjects;
    const rt_material* Materials; int64_t nMaterials;
    rt_triangle* Triangles; int64_t nTriangles; /* written by orc_calc_tri_normals */
    const rt_vertex* Vertices; int64_t nVertices;
    const rt_float4* TextureAtlas; int32_t atlasW, atlasH;
    const rt_float4* EnvironmentMap; int32_t envW, envH;
} orc_scene;

/* Optional per-pixel trace record (not part of the reference's outputs): what the north star requires to be
 * bit-exact.  Filled by orc_initial_trace (primary + shadow ray) and orc_trace_rays (bounce rays). */
typedef struct {
    int32_t triIndex;      /* closest-hit triangle of the primary ray, -1 on a miss */
    int32_t objIndex;      /* its scene object, -1 on a miss */
    uint32_t visitHash;    /* FNV-style hash over every BVH node popped by the primary ray, in order */
    int32_t lightObj;      /* LightObjectReservoirSample result, -1 if not reached */
    int32_t lightTri;      /* TriReservoirSample result, -1 if not reached */
    uint32_t shadowVisitHash; /* same hash for the VIS_REUSE shadow ray, 0 if none */
    int32_t shadowTri;     /* closest-hit triangle of the shadow ray, -1 if none/miss */
    uint32_t bounceVisitHash; /* same hash over all rays cast by TraceRays for the pixel, 0 if none */
} orc_trace_record;

/* Whole-dispatch counters (summed over all pixels); the inputs of the algorithmic-bytes formula. */
typedef struct {
    uint64_t rays;        /* rays cast (GetRayCollision calls) */
    uint64_t pops;        /* nodes popped in both traversal loops (ReSTIR.compute:171,227) */
    uint64_t bvChecks;    /* child boxes fetched and tested (debugData.bvChecks) */
    uint64_t objVisits;   /* scene-object leaf iterations (:237) */
    uint64_t triIters;    /* leaf triangle-loop iterations (:181) */
    uint64_t triChecks;   /* triangles passing the back-face pre-cull (debugData.triChecks) */
    uint64_t validNeighbours; /* SpatialReusePass: neighbours whose reservoir was fetched (:534) */
    uint64_t activePixels;    /* pixels not early-returned */
    uint64_t temporalCopies;  /* TemporalReuse: pixels whose reprojection missed, copy-and-return (:600-604) */
    uint64_t temporalMerges;  /* TemporalReuse: pixels merged with their history (:617 true) */
    uint64_t temporalRejects; /* TemporalReuse: pixels whose hit point moved too far (:617 false) */
} orc_counters;

typedef struct {
    rt_hit_info* HitInfos;
    rt_reservoir* Candidates;
    rt_reservoir* CandidatesB;
    rt_reservoir* TemporalFrameBuffer;
    rt_float4* Result;
    rt_float4* AccumulatedResult;
    rt_float4* DebugOverlay;
    float* DepthBuffer;
    rt_float4* NormalsBuffer;  /* float3 view of an RGBA target; w is written as 0 */
    rt_float4* RayHitPointsA;
    rt_float4* RayHitPointsB;
    orc_trace_record* records; /* may be NULL */
    orc_counters* counters;    /* may be NULL; accumulated into */
} orc_frame;

/* Each returns 0 on success. `threads` <= 0 means all available cores. Rows are split over threads in 8x8 tiles. */
int orc_calc_tri_normals(orc_scene* scene, int threads);                                   /* PreCalc.compute:39-50 */
int orc_initial_trace(const orc_scene*, const orc_params*, orc_frame*, int threads);       /* ReSTIR.compute:463-510 */
int orc_temporal_reuse(const orc_scene*, const orc_params*, orc_frame*, int threads);      /* :586-641 */
int orc_spatial_reuse_pass(const orc_scene*, const orc_params*, orc_frame*, int threads);  /* :558-571 */
int orc_transfer_to_original(const orc_scene*, const orc_params*, orc_frame*, int threads);/* :575-582 */
int orc_trace_rays(const orc_scene*, const orc_params*, orc_frame*, int threads);          /* :759-815 */
int orc_accumulate_frames(const orc_params*, orc_frame*, int frameCount, int threads);     /* PostProcessor.compute:10-19 */

/* Known-answer helpers exposed for tests. */
uint32_t orc_next_random(uint32_t* state);                       /* RendererResources.hlsl:63-69 */
float orc_rand_normalized(uint32_t* state);                      /* :71-74 */
int32_t orc_rand_int_span(int32_t a, int32_t b, uint32_t* state);/* :76-83 */
int32_t orc_weighted_rand(float a, float b, uint32_t* state);    /* :85-91 */
float orc_ray_bv_intersect(const float pos[3], const float dir[3], const rt_bv* bv); /* ReSTIR.compute:148-161 */
/* ReSTIR.compute:119-146; returns didHit, writes dst and uv. */
int32_t orc_ray_triangle_intersect(const float pos[3], const float dir[3], const rt_vertex* v0, const rt_vertex* v1,
                                   const rt_vertex* v2, float* dst, float uv[2]);
float orc_get_brdf(const float incoming[3], const float outgoing[3], const float normal[3], float metallicity,
                   float roughness);                             /* BRDF.hlsl:25-65 */
/* det-math passthroughs (include/rt_detmath.h) so tests can compare them with libm. which: 0 sin,1 cos,2 log,
 * 3 exp,4 asin,5 acos,6 atan; pow and atan2 take two arguments. */
float orc_detmath1(int which, float x);
float orc_detmath2(int which, float x, float y);
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
