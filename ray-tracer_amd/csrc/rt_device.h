// rt_device.h — data structures shared by the host side of librtnative (rt_native.hip) and its gfx950 kernels.
//
// Internal to the library; nothing here crosses the C-ABI.  The ABI-level layouts are in include/rt_types.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_types.h"

namespace rtn {

// ---- internal memory layouts (HBM) -------------------------------------------------------------------------
//
// Wide BVH node, 64 bytes, one per BV index (only the slots of internal nodes are filled or ever read).
// It holds what one traversal step needs — both children's boxes and both children's stack entries — so an
// internal visit is ONE aligned 64-byte fetch instead of the reference's three dependent 32-byte BV fetches
// (BVH_ReSTIR_DL/.../ReSTIR_BVH_RayTracer.compute:171,198-199).
//   q0 = minA.xy, maxA.xy   q1 = minB.xy, maxB.xy   q2 = minA.z, minB.z, maxA.z, maxB.z   q3 = entryA, entryB, 0, 0
// (x and y of a box together, z of both boxes together: the box test then runs on register pairs, rt_kernels.hip)
//
// Stack entry (one dword):
//   bit31 = 1            leaf, components [start, start+count): count = bits 26..30, start = bits 0..25
//   bit31 = 0, bit30 = 1 leaf whose range does not fit the inline form; low 30 bits = BV index to read it from
//   bit31 = 0, bit30 = 0 internal node; low 30 bits = BV index (= index of its wide record)
constexpr uint32_t ENTRY_LEAF = 0x80000000u;
constexpr uint32_t ENTRY_LEAF_INDIRECT = 0x40000000u;
constexpr uint32_t ENTRY_INDEX_MASK = 0x3fffffffu;
// Top of the tree in LDS (an alternative that is measured, not the default: rt_set_tuning "top_nodes_lds").  The first
// TOP_MAX internal nodes in breadth-first order from the scene root - through the scene-level leaves into the mesh BVHs
// of the objects - have a copy of their wide record in a small table that every wave of a launch loads into LDS.  An
// internal entry with ENTRY_TOP set names a slot of that table instead of a BV index (BV indices stay below 2^26).  Only
// records of the table, the scene root entry and the objects' root entries ever hold such an entry (a node's parent comes
// before it in breadth-first order), so the wide array itself and the kernels that do not use the table are untouched.
constexpr uint32_t ENTRY_TOP = 0x20000000u;
constexpr int TOP_MAX = 64;
struct TopPlan {
    uint32_t node[TOP_MAX];             // slot -> BV index
    int16_t slotA[TOP_MAX], slotB[TOP_MAX];   // slot of each child, -1 when the child is not in the table (or a leaf)
    int n;
};
constexpr int ENTRY_INLINE_MAX_COUNT = 31;
constexpr int ENTRY_INLINE_MAX_START = (1 << 26) - 1;

// Baked triangle, 64 bytes, one per Triangle: everything the leaf loop needs in one aligned fetch instead of a
// 28-byte Triangle plus three dependent 20-byte Vertex gathers (:121-123,181).
//   q0 = v0.xyz, N.x   q1 = edgeAB.xyz, N.y   q2 = edgeAC.xyz, N.z   q3 = localNormal.xyz, 0
// with N = cross(edgeAB, edgeAC), computed with the same fp32 operations the intersection routine would use.

// Per-pixel records that are only ever accessed at the thread's own pixel (HitInfos, TemporalFrameBuffer) are kept
// as planes of 16-byte elements (structure of arrays): the 52-byte record of `count` pixels is stored as three float4
// planes followed by one float plane, in the same 52*count bytes.  Every access is then one coalesced
// 16-byte-per-lane load or store.
//   hit info : p0 = hitPoint.xyz, dst       p1 = normal.xyz, materialIndex  p2 = incomingDir.xyz, uv.x       p3 = uv.y
//   reservoir: p0 = dir.xyz, chosenWeight   p1 = hitPoint.xyz, totWeights   p2 = normal.xyz, totCandidates   p3 = brdf
struct Planes {
    float4* p0;
    float4* p1;
    float4* p2;
    float* p3;
    int64_t count;
};
// Candidates and CandidatesB are also GATHERED, from random neighbouring pixels (SpatialReusePass reads nine per pixel
// and pass): there a reservoir is one aligned 64-byte record, so a gathered one costs one cache line instead of four.
//   q0 = dir.xyz, chosenWeight   q1 = hitPoint.xyz, totWeights   q2 = normal.xyz, totCandidates   q3 = brdf, 0, 0, 0
// When the library keeps two copies of such a buffer (one per frame in flight, rt_native.hip), every copy has a tag per
// pixel: the number (rt_ctx_s::epoch, never 0) of the frame that last wrote the pixel's record INTO THAT COPY, stored
// with every record.  Only the frames that work on a copy write its tags, so two frames in flight never write the same
// tag.  A readback takes each pixel from the copy with the larger tag: pixels a frame does not write (no hit) still
// read back what an earlier frame left, as with a single buffer.
struct Recs {
    float4* q;  // 4 per pixel
    int64_t count;
    uint32_t* tag;      // null: a single copy
    uint32_t tagValue;  // the frame's number
};

// The other copies of such a buffer, for a readback (at most three: four frames in flight).
struct OtherCopies {
    const float4* q[3];
    const uint32_t* tag[3];
    int n;
};

struct Tex {
    float4* rgba;  // RGBA textures (fp32 storage)
    float* r;      // R32F textures
    int w, h;
    int unorm8;    // RT_FORMAT_RGBA8_UNORM render target: stores clamp to [0, 1] and round to a multiple of 1/255
};

// Mirror of the shader's constants (ReSTIR_BVH_RayTracer.compute:14-66,91-92), snapshotted at dispatch time from
// the typeless constant table, plus the resolved buffer bindings of the dispatch.  Passed to kernels by value.
struct KParams {
    // scene
    const float4* wide;       // wide BVH nodes, 4 x float4 per BV index
    const float4* btri;       // baked triangles, 4 x float4 per triangle
    const rt_bv* BVs; int64_t nBVs;
    const rt_scene_object* objs; int64_t nObjs;
    const uint32_t* objRootEntry;  // stack entry of each scene object's mesh-BVH root
    const rt_light_object* lights; int64_t nLights;
    const rt_material* mats; int64_t nMats;
    const rt_triangle* tris; int64_t nTris;
    const rt_vertex* verts; int64_t nVerts;
    uint32_t sceneRootEntry;
    int stackDepth;  // dwords of LDS stack per lane
    // top of the tree in LDS (see ENTRY_TOP): the table, its slots in use (0: none), and the entries that lead into it
    const float4* top; int topCount;
    const uint32_t* objRootEntryTop;
    uint32_t sceneRootEntryTop;
    Tex atlas, env;
    // per-pixel state
    Planes hit, temporal;
    Recs cand, candB;
    Tex result, accumulated, debugOverlay, depth, normals, hitA, hitB;
    // constants
    uint32_t MaxBVHDepth;
    float ScatterProbability;
    uint32_t MaxBounceCount;
    int32_t ResX, ResY;
    float ViewSpaceDimsX, ViewSpaceDimsY, DefocusStrength, FocalPlaneFactor;
    float camPos[3], lastCamPos[3];
    float camT[9], lastCamTInv[9];  // row-major 3x3
    uint32_t FrameRand, FrameCount;
    int32_t DebugMaxX, DebugMaxY;
    uint32_t SceneObjectReservoirTestsNum, TriReservoirTestsNum;
    int32_t ReuseBufferCycle, MaxOffset;
    uint32_t TemporalCandidatesNum;
    float SpatialHitPointDiffThreshold, SpatialNormalsAngleThreshold, SpatialBRDFThreshold;
    float TemporalReuseWeight, PixelMovementThreshold, TemporalPrecisionThreshold;
    uint32_t EmittingObjectsNum;
    float atlasDimsX, atlasDimsY, envDimsX, envDimsY;  // raw constant words reinterpreted as float
    int32_t kwVisReuse, kwWeightRecalc, kwBrdf;
    int32_t ppFrameCount;
    // RIS project (BVH_RIS_DL .../RIS_BVH_RayTracer.compute:6-59): its own record types and constants
    const rt_ris_tri* risTris;
    const rt_ris_material* risMats;
    rt_ris_hit_info* risHit; int64_t nRisHit;
    float4* risCand; int64_t nRisCand;
    uint32_t* risTag; uint32_t risTagValue;   // per pixel ray, when the library keeps several copies of Candidates (see Recs)
    Tex rayHitPoints;
    int32_t RaysPerPixel, SceneObjectCandidatesNum, TriCandidatesNum, TotCandidatesNum, CandidateReservoirTestsNum;
    int32_t risMaxBounceCount, risMaxBVHDepth, risFrameRand;
    int32_t risSceneObjectReservoirTestsNum, risTriReservoirTestsNum, risEmittingObjectsNum;
    // band + instrumentation
    int32_t row0, row1;
    const uint32_t* blockMap;  // launch block index -> (by << 16 | bx) of the 16x16-pixel block it renders, ~0u = none
    uint32_t* blockCost;       // InitialTrace only, may be null: hit pixels per wave, [(by * blocksX + bx) * 4 + wave]
    uint32_t* blockSteps;      // TraceRays only, may be null: steps each wave needed for its rays, same indexing
    int blocksX;               // blocks per image row of this launch's grid
    uint32_t* bgCounter;       // background primary-ray launch only: next table round of each XCD (8 counters, zeroed per launch)
    uint32_t bgBlocks;         // ... and the number of table entries
    unsigned long long* stats;  // sharded counters (rt_dispatch_stats order) or null
    int32_t statsLevel;         // 1: rays only, 2: everything
    uint32_t* records;          // 8 dwords per pixel (trace record) or null
    unsigned long long* waveProfile;  // diagnostics: 4 qwords per wave of the launch (start, end, max iterations, sum iterations) or null
    int32_t waveProfileKernel;        // which kernel writes it: 0 primary, 1 candidates, 2 trace_rays
};

// Trace record written when a "TraceRecords" buffer is bound (8 dwords per pixel):
//   0 triIndex  1 objIndex  2 visitHash  3 lightObj  4 lightTri  5 shadowVisitHash  6 shadowTri  7 bounceVisitHash
constexpr int RECORD_DWORDS = 8;

}  // namespace rtn
