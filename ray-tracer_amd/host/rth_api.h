/* rth_api.h — C interface of librthost.so: the host side that sits ABOVE the C-ABI of librtnative.so.
 *
 * In the reference the host side is Unity C#.  This library restates, in C++, the parts of it that feed and
 * drive the compute path, so the path can run and be measured headless:
 *   - the scene/BVH builder  (BVH_ReSTIR_DL/Assets/Scripts/ObjectHelper.cs, Resources.cs, Data types/Other/)
 *   - the frame driver       (BVH_ReSTIR_DL/Assets/Scripts/Main.cs, ShaderHelper.cs, ComputeHelper.cs)
 * The frame driver only ever talks to the GPU through include/rt_native.h, exactly as the C# would through
 * P/Invoke.  Python (tests, bench.py) binds this interface with ctypes.
 */
#ifndef RTH_API_H
#define RTH_API_H

#include "../../include/rt_native.h"
#include "../../include/rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------ meshes ------------------------------------------------ */
typedef struct rth_mesh_s* rth_mesh;

rth_mesh rth_mesh_create(const rt_vertex* verts, int nverts, const int32_t* tris, int ntris);
/* Loads the .rtmesh format written by tools/import_obj.py.  NULL on failure. */
rth_mesh rth_mesh_load(const char* path);
/* Splits every triangle into 9 by trisecting its edges (shared edges share the new vertices).  New mesh. */
rth_mesh rth_mesh_subdivide9(rth_mesh m);
/* Returns a copy whose vertex positions are displaced by a seeded pseudo-random offset of at most `amplitude`
 * per axis, so that otherwise identical meshes are distinct geometry. */
rth_mesh rth_mesh_jitter(rth_mesh m, uint32_t seed, float amplitude);
/* Procedural fan of `ntris` triangles around the z axis (unit radius), facing +z.  Used for emissive quads/discs. */
rth_mesh rth_mesh_disc(int ntris);
/* The same disc as sectors x rings of similar-sized triangles (ntris = sectors * (2 * rings - 1)). */
rth_mesh rth_mesh_ring_disc(int ntris);
void rth_mesh_free(rth_mesh m);
int rth_mesh_counts(rth_mesh m, int* nverts, int* ntris);
const rt_vertex* rth_mesh_vertices(rth_mesh m);
const int32_t* rth_mesh_triangles(rth_mesh m);

/* ------------------------------------------------ scene ------------------------------------------------- */
typedef struct rth_scene_s* rth_scene;

/* MaterialInput (Data types/Other/MaterialInput.cs) without textures: the scalar path of ConstructTextureAtlas
 * (ObjectHelper.cs:390-449). */
typedef struct {
    float brightness;
    float col[3];
    float roughness;
    float metallicity;
    float bump;
} rth_material_input;

/* A texture of a MaterialInput: width x height RGBA8 texels, row 0 first (copied).  pixels == NULL: no texture. */
typedef struct {
    int32_t width, height;
    const unsigned char* pixels;
} rth_texture_input;
enum { RTH_TEX_COL = 0, RTH_TEX_BUMP = 1, RTH_TEX_ROUGHNESS = 2, RTH_TEX_METALLICITY = 3, RTH_TEX_NORMALS = 4 };

/* splitResolution, maxDepthSceneBVH: ObjectHelper.SplitResolution / MaxDepthSceneBVH (ObjectHelper.cs:19-20). */
rth_scene rth_scene_create(int splitResolution, int maxDepthSceneBVH);
void rth_scene_free(rth_scene s);
/* Returns the material index. */
int rth_scene_add_material(rth_scene s, const rth_material_input* m);
/* The same with the material's five optional textures (indexed by RTH_TEX_*): ConstructTextureAtlas
 * (ObjectHelper.cs:370-452) packs them into one RGBA8 atlas when the scene is built and the material then refers to
 * its textures by location and size in the atlas (col / bump / roughness / metallicity = -1 select the texture).
 * Unity's Texture2D.PackTextures is replaced by a deterministic shelf packer with the same 1-texel padding, so the
 * locations differ from Unity's; the kernels only ever see locations.  As in the reference, metallicity textures are
 * NOT added to the list that is packed although they consume a rectangle (:373-379 against :430-437): a material
 * with one is given the rectangle of the texture that follows, the last texture of the scene finds none left, and
 * rth_scene_build fails where the C# throws IndexOutOfRangeException. */
int rth_scene_add_material_textured(rth_scene s, const rth_material_input* m, const rth_texture_input tex[5]);
/* Registers a mesh (copied) and returns its mesh id.  Objects that share a mesh id share one BVH, which is
 * what the reference's mesh-key lookup does (ObjectHelper.cs:582-594). */
int rth_scene_add_mesh(rth_scene s, rth_mesh m);
/* A GameObject: transform (position, Euler angles in degrees, scale) + SceneObjectSettings.{MaterialIndex,MaxDepthBVH}. */
int rth_scene_add_object(rth_scene s, int meshId, const float position[3], const float eulerDeg[3],
                         const float scale[3], int materialIndex, int maxDepthBVH);
/* ObjectHelper.ConstructScene (ObjectHelper.cs:666-738) on the registered objects.  threads <= 0: all cores. */
int rth_scene_build(rth_scene s, int threads);
/* transform.position / rotation / localScale of object `objectIndex` (the value rth_scene_add_object returned) change.
 * In the reference that makes SceneObjectSettings.LateUpdate request a reload (SceneObjectSettings.cs:27-36). */
int rth_scene_set_object_transform(rth_scene s, int objectIndex, const float position[3], const float eulerDeg[3], const float scale[3]);
/* ConstructScene called again on a built scene (what Main.SetData does on a reload, Main.cs:278): meshes already loaded
 * keep their BVHs, vertices and triangles (ObjectHelper.cs:582-594); the per-object matrices and boxes, the scene BVH
 * and the light list are made anew.  keepOldSceneBVH = 1 is the ReSTIR project's behaviour: the new scene BVH is
 * appended behind the old one, which stays in BVs unreferenced (LastSceneBVHLength is never set, ObjectHelper.cs:38,
 * 613-614), so BVsNum grows and SceneBVHStartIndex moves up with every reload; 0 is the RIS project's, which drops the
 * old scene BVH first (BVH_RIS_DL/Assets/Scripts/MeshHelper.cs:371-372,418).  Pointers obtained from rth_scene_get are
 * invalidated. */
int rth_scene_rebuild(rth_scene s, int keepOldSceneBVH);

typedef struct {
    const rt_bv* BVs; int64_t nBVs;
    const rt_vertex* Vertices; int64_t nVertices;
    const rt_triangle* Triangles; int64_t nTriangles;   /* RenderTriangle[] in BVH order, localNormal = 0 */
    const rt_scene_object* SceneObjects; int64_t nSceneObjects;
    const rt_light_object* LightObjects; int64_t nLightObjects;
    const rt_material* Materials; int64_t nMaterials;
    const rt_int2* ComponentDatas; int64_t nComponentDatas; /* (componentStart, totComponents) per BV node */
    int32_t maxBVHDepth, emittingObjectsNum, sceneBVHStartIndex;
    float totArea;
    const unsigned char* atlas; int32_t atlasWidth, atlasHeight; /* RGBA8, row 0 first; 1x1 black without textures */
} rth_scene_data;
int rth_scene_get(rth_scene s, rth_scene_data* out);

/* Scene cache, JSON multi-file form (FileLoader.cs:12-91): writes `<folder>/<key>.json` for loadedBVs, loadedVertices,
 * renderTriangles, sceneObjectDatas, lightObjects, loadedComponentDatas, integers, floats (and an empty renderBVs, as
 * the reference's SaveToFile leaves it, ObjectHelper.cs:523-544).  The folder must exist. */
int rth_scene_save_json(rth_scene s, const char* folder);
/* Fills `s` from such a folder instead of building it (ConstructScene in LoadExistingFile mode,
 * ObjectHelper.cs:672-677): the scene needs its materials registered (they are not part of the cache) but no
 * meshes or objects.  Each file is looked up as `<key>.json` and, if `prefix` is given, `<prefix>_<key>.json`.
 * On failure returns -1 and writes a message naming the file into `err`. */
int rth_scene_load_json(rth_scene s, const char* folder, const char* prefix, char* err, int errLen);

/* Utils.CreateWorldToLocalMatrix (Resources.cs:67-87) and its inverse, column-major. */
void rth_world_to_local(const float position[3], const float eulerDeg[3], const float scale[3], float outW2L[16],
                        float outL2W[16]);

/* --------------------------------------------- frame driver --------------------------------------------- */
/* ViewSpaceDims as Main.UpdateSettings computes it (Main.cs:223-228; the aspect ratio is an integer division). */
void rth_view_space_dims(float fieldOfView, int resX, int resY, float out[2]);
/* Camera localToWorld matrix (column-major) from position + rotation quaternion, and its inverse (Main.cs:194-199). */
void rth_camera_matrices(const float position[3], const float rotation[4], float outTransform[16], float outInverse[16]);

typedef struct rth_main_s* rth_main;

/* The inspector fields of Main (Main.cs:8-53) that reach the compute path, plus the camera transform. */
typedef struct {
    int32_t Resolution[2];
    float fieldOfView;
    int32_t MaxBounceCount;
    float ScatterProbability, DefocusStrength, FocalPlaneFactor;
    int32_t DebugMaxTriChecks, DebugMaxBVChecks;
    int32_t SceneObjectReservoirTestsNum, TriReservoirTestsNum, CandidateReservoirTestsNum;
    int32_t SpatialReuseIterations, TemporalCandidatesNum, SpatialReuseBlur;
    float TemporalPrecisionThreshold, TemporalReuseWeight, PixelMovementThreshold;
    float SpatialHitPointDiffThreshold, SpatialNormalsAngleThreshold, SpatialBRDFThreshold, VisibilityReuseThreshold;
    int32_t DoVisibilityReuse, DoWeightRecalc, DoBRDF;
    float cameraPosition[3];
    float cameraRotation[4]; /* quaternion x,y,z,w */
    /* FrameRand for frame k is (frameRandBase + frameRandStep*k) mod 999999; the reference draws
     * UnityEngine.Random.Range(0, 999999) (Main.cs:137), which is not reproducible. */
    uint32_t frameRandBase, frameRandStep;
    /* 0: the 3- and 4-channel render targets are created as fp32 (results comparable to full precision);
     * 1: as RT_FORMAT_RGBA8_UNORM, the format TextureHelper.CreateTexture gives them (TextureHelper.cs:144-155). */
    int32_t Unorm8RenderTargets;
} rth_main_config;

void rth_main_default_config(rth_main_config* cfg); /* values of the committed Unity scene (Ray Tracer.unity:622-652) */

/* Main.Start (Main.cs:105-115): opens the three shaders on `ctx`, uploads the scene (SetData, :273-310),
 * pushes all settings (UpdateSettings, :211-271).  The scene must be built.  row0/row1: image rows this context
 * renders (row1 <= 0: all rows). */
rth_main rth_main_create(rt_ctx ctx, const rth_main_config* cfg, rth_scene scene, int row0, int row1);
void rth_main_destroy(rth_main m);
const char* rth_main_last_error(rth_main m);

/* One Unity frame: Update (Main.cs:117-132: UpdatePerFrame :134-145, then UpdateSettings if a reload was requested),
 * then OnRenderImage -> RunReSTIRShader (:404-415) and RunPostProcessingShader (:417-420).  Asynchronous; returns
 * after enqueueing. */
int rth_main_render_frame(rth_main m);
/* Requests what OnValidate / SceneObjectSettings.LateUpdate request (Main.cs:206-209, SceneObjectSettings.cs:27-36):
 * DoUpdateSettings and DoReloadData.  The NEXT Update then runs UpdateSettings(true) after its UpdatePerFrame
 * (Main.cs:127-129): FrameCount restarts at 0 from the frame after, SetData releases all ten buffers, takes the scene
 * arrays from the scene again (call rth_scene_rebuild first if objects moved), uploads them, runs CalcTriNormals and
 * creates the four per-pixel buffers anew (zeroed), and every setting is pushed again.  The render targets stay as
 * they are.  cfg: new inspector values, or NULL to keep them (Resolution and Unorm8RenderTargets may not change). */
int rth_main_reload(rth_main m, const rth_main_config* cfg);
/* The two halves separately, for tests that interleave readbacks (rth_main_update_per_frame is Update: it performs
 * a requested reload): */
int rth_main_update_per_frame(rth_main m);
int rth_main_run_restir_shader(rth_main m);
int rth_main_run_post_processing(rth_main m);
/* Individual dispatches of RunReSTIRShader, for per-kernel parity tests and timing. stage: 0 InitialTrace,
 * 1 TemporalReuse, 2 SpatialReuse (all passes + transfer), 3 TraceRays. */
int rth_main_run_stage(rth_main m, int stage);
/* Spatial pass i alone (MaxOffset = 3^i * SpatialReuseBlur, ReuseBufferCycle = i even), and the TransferToOriginal
 * that follows an odd number of passes: a multi-GPU host exchanges reservoir halos between these. */
int rth_main_run_spatial_pass(rth_main m, int i);
int rth_main_run_transfer(rth_main m);
int rth_main_frame_count(rth_main m);
/* Moves the camera (transform.position / transform.rotation, quaternion x,y,z,w): the next UpdatePerFrame pushes the
 * new WorldSpaceCameraPos / CameraTransform and the previous frame's as LastWorldSpaceCameraPos /
 * LastCameraTransformInverse, as SetCameraData does every frame (Main.cs:184-204). */
int rth_main_set_camera(rth_main m, const float position[3], const float rotation[4]);

/* Named access to the driver's buffers and textures ("Candidates", "CandidatesB", "TemporalFrameBuffer",
 * "HitInfos", "Triangles", ...; "Result", "AccumulatedResult", "DebugOverlay", "DepthBuffer", "NormalsBuffer",
 * "RayHitPointsA", "RayHitPointsB").  NULL if unknown. */
rt_buffer rth_main_buffer(rth_main m, const char* name);
rt_texture rth_main_texture(rth_main m, const char* name);
rt_shader rth_main_shader(rth_main m, const char* name); /* "rt", "pp", "pc" */

/* ------------------------------------------- multi-GPU frame loop --------------------------------------- */
/* One process per GPU; the image is cut into row bands (rt_set_row_band), the scene replicated, reservoir/depth
 * halos exchanged once per frame and the displayed image completed on rank 0 with RCCL point-to-point transfers on
 * the context's transfer stream (rt_transfer_stream), beside the kernels: the halos travel while the first spatial
 * pass runs on the rows that need none, the gather of frame f while frame f+1 traces its primary rays.  No
 * counterpart in the reference, which is single-GPU. */
typedef struct rth_mg_s* rth_mg;
/* Band boundaries b[0..world] (multiples of 8 rows except the last).  -1 if a band would be empty. */
int rth_mg_band_bounds(int height, int world, int* out);
/* The same from a cost per row: bands of about equal cost, none shorter than minRows. */
int rth_mg_balanced_bounds(int height, int world, const double* rowCost, int minRows, int* out);
/* halo[i] / extend[i] per spatial pass (see the definition); returns the rows exchanged per frame. */
int rth_mg_halo_plan(int iterations, int blur, int* halo, int* extend);
/* ncclGetUniqueId on the calling rank; the 128 bytes must reach every rank (any side channel) before rth_mg_create. */
int rth_mg_unique_id(unsigned char out[128]);
/* bounds: band boundaries b[0..world] every rank agrees on, or NULL for rth_mg_band_bounds' equal heights.
 * Collective over all ranks (ncclCommInitRank) when an id is given; without one no communicator is made (fine for
 * world == 1; for world > 1 the rank can then only be stepped with rth_mg_phase / rth_mg_plan).  `m` must have
 * been created on a context of this rank's GPU with full-frame buffers (row0 = row1 = 0). */
rth_mg rth_mg_create(rth_main m, int rank, int world, const int* bounds, const unsigned char uniqueId[128], char* err, int errLen);
void rth_mg_destroy(rth_mg g);
const char* rth_mg_last_error(rth_mg g);
int rth_mg_bounds(rth_mg g, int* out);
/* New cuts (before the first frame, the same on every rank). */
int rth_mg_set_bounds(rth_mg g, const int* bounds);
/* Cost-balanced cuts.  rth_mg_row_costs: this rank traces the primary rays of frame 0 on its present band (the real
 * frame 0 repeats them; the frame counters are put back) and writes a cost per row for its own rows into rowCost[H],
 * 0 elsewhere.  rth_mg_balance (collective, needs a communicator): sums those vectors over the ranks with one
 * ncclAllReduce and sets the cuts rth_mg_balanced_bounds derives from the sum.  Call before the first frame. */
int rth_mg_row_costs(rth_mg g, double* rowCost);
int rth_mg_balance(rth_mg g);
/* Self-test: moves rows [src0, src1) of Candidates and DepthBuffer to row dst0 through RCCL send/recv addressed to
 * this same rank (needs a communicator, i.e. a unique id at creation). */
int rth_mg_loopback(rth_mg g, int src0, int src1, int dst0);
/* One transfer of a frame's exchange: bytes at ptr (device memory of this rank) sent to, or received from, peer. */
typedef struct { int32_t send; int32_t peer; void* ptr; size_t bytes; } rth_xfer;
/* A frame in steps, for tests that hold several ranks in one process and move the data themselves.
 * Phases: 0 = InitialTrace and TemporalReuse on the band; 1 = the first spatial pass on the rows that need no halo;
 * 2 = the rest of the spatial reuse (needs the halos); 3 = TraceRays and AccumulateFrames.
 * Plans: 0 = the halo exchange (may be carried out any time between phase 0 and phase 2), 1 = the gather of the
 * displayed image (any time between phase 3 and the next frame's phase 3), as the lists of transfers
 * rth_mg_render_frames hands to RCCL, in order.  rth_mg_plan returns the number of entries, or -1. */
int rth_mg_phase(rth_mg g, int phase);
int rth_mg_plan(rth_mg g, int which, rth_xfer* out, int cap);
/* Enqueues n banded frames (asynchronous).  After rt_sync, rank 0's Result texture holds the whole image.  Collective:
 * every rank calls it with the same n, and with the same timing switch (rth_mg_enable_timing), because the order of
 * the transfers depends on both. */
int rth_mg_render_frames(rth_mg g, int n);
/* Rehearsal on ONE GPU of what rth_mg_render_frames does on several: the ranks are threads of one process, each with a
 * context of its own on the same GPU (which RCCL refuses), and their transfers meet in a hub - same matching and
 * ordering rules as RCCL's grouped send / receive, carried out as device copies.  Every rank's thread calls
 * rth_mg_render_frames as usual.  rth_mg_balance and rth_mg_loopback are not available on such a rank. */
typedef struct rth_hub_s* rth_hub;
rth_hub rth_hub_create(int world);
void rth_hub_destroy(rth_hub hub);   /* after the ranks made with it */
int rth_mg_use_hub(rth_mg g, rth_hub hub);   /* `g`: made by rth_mg_create / rth_mg_create_ris without a communicator id */
/* Per-frame breakdown, averaged over the frames rendered since timing was switched on (each frame is then waited
 * for, so frames no longer overlap): milliseconds on the dispatch stream for the whole frame, before the halo
 * exchange, for the interior rows of the first spatial pass and for everything after; on the transfer stream for the
 * halo exchange and the gather; and the part of the halo exchange the dispatch stream had to wait for. */
enum { RTH_MG_T_FRAME = 0, RTH_MG_T_BEFORE_HALO, RTH_MG_T_INTERIOR, RTH_MG_T_AFTER_HALO, RTH_MG_T_HALO, RTH_MG_T_HALO_EXPOSED,
       RTH_MG_T_GATHER, RTH_MG_TIMES };
int rth_mg_enable_timing(rth_mg g, int on);
int rth_mg_times(rth_mg g, double out[RTH_MG_TIMES]);   /* returns the number of frames averaged */

/* ------------------------------------- frame driver of the RIS project ---------------------------------- */
typedef struct rth_ris_main_s* rth_ris_main;

/* The inspector fields of the RIS project's Main (BVH_RIS_DL/Assets/Scripts/Main.cs:11-38) that reach the compute
 * path, the camera transform, and the one material input its Material2 adds (MatTypesInput2.x). */
typedef struct {
    int32_t Resolution[2];
    float fieldOfView;
    int32_t MaxBounceCount, RaysPerPixel;
    float ScatterProbability, DefocusStrength, FocalPlaneFactor;
    int32_t DebugMaxTriChecks, DebugMaxBVChecks;
    int32_t SceneObjectCandidatesNum, TriCandidatesNum;
    int32_t SceneObjectReservoirTestsNum, TriReservoirTestsNum, CandidateReservoirTestsNum;
    float cameraPosition[3];
    float cameraEulerDeg[3]; /* transform.rotation.eulerAngles (Main.cs:190) */
    float smoothness;        /* Material2.smoothness of every material */
    uint32_t frameRandBase, frameRandStep; /* as in rth_main_config */
    int32_t Unorm8RenderTargets; /* 1: render targets as `new RenderTexture(w, h, 24)` makes them (Main.cs:324-364): 8-bit UNORM */
} rth_ris_config;

/* CameraTransform as Main.SetCameraOrientationAndTransform builds it from the Euler angles (Main.cs:189-211), column-major. */
void rth_ris_camera_transform(const float eulerDeg[3], float out[16]);
void rth_ris_default_config(rth_ris_config* cfg); /* BVH_RIS_DL/Assets/Scenes/Ray Tracer.unity:381-426 */

/* Main.Start (Main.cs:87-103) on a built scene: derives the RIS project's triangle and material arrays from it,
 * opens RIS_BVH_RayTracer / PostProcesser / PreCalc on `ctx`, uploads everything and runs CalcTriNormals. */
rth_ris_main rth_ris_main_create(rt_ctx ctx, const rth_ris_config* cfg, rth_scene scene, int row0, int row1);
void rth_ris_main_destroy(rth_ris_main m);
const char* rth_ris_main_last_error(rth_ris_main m);
/* One frame: UpdatePerFrame (:132-144), RunRenderShader (:319-371), RunPostProcessingShader (:373-376). */
int rth_ris_main_render_frame(rth_ris_main m);
int rth_ris_main_update_per_frame(rth_ris_main m);   /* Update: performs a requested reload after UpdatePerFrame */
/* DoUpdateSettings (BVH_RIS_DL Main.cs:117,126,218; SceneObjectSettings.cs:26,32; any camera or object move sets it):
 * the next Update runs UpdateSettings -> SetData after its UpdatePerFrame: FrameCount restarts, all seven buffers are
 * released and created again from the scene as it is now, CalcTriNormals runs again. */
int rth_ris_main_reload(rth_ris_main m);
int rth_ris_main_run_stage(rth_ris_main m, int stage); /* 0 GenerateCandidates, 1 TraceRays */
int rth_ris_main_run_post_processing(rth_ris_main m);
int rth_ris_main_frame_count(rth_ris_main m);
rt_ctx rth_ris_main_ctx(rth_ris_main m);
int rth_ris_main_resolution(rth_ris_main m, int out[2]);
/* One rank of the banded RIS frame loop: the RIS kernels read only their own pixel, so a frame is the rank's band of
 * GenerateCandidates, TraceRays and AccumulateFrames followed by the gather of Result on rank 0; no halo.  The
 * handle is used with the same rth_mg_* calls as the ReSTIR one (rth_mg_loopback excepted). */
rth_mg rth_mg_create_ris(rth_ris_main m, int rank, int world, const int* bounds, const unsigned char uniqueId[128], char* err, int errLen);

typedef struct {
    const rt_ris_tri* Tris; int64_t nTris;               /* worldNormal = 0: CalcTriNormals fills it on the device */
    const rt_ris_material* Materials; int64_t nMaterials;
} rth_ris_scene_data;
int rth_ris_main_scene(rth_ris_main m, rth_ris_scene_data* out);
rt_buffer rth_ris_main_buffer(rth_ris_main m, const char* name);   /* "Tris", "Candidates", "HitInfos", ... */
rt_texture rth_ris_main_texture(rth_ris_main m, const char* name); /* "Result", "AccumulatedResult", "DebugOverlay", "RayHitPoints" */
rt_shader rth_ris_main_shader(rth_ris_main m, const char* name);   /* "rt", "pp", "pc" */

#ifdef __cplusplus
}
#endif
#endif
