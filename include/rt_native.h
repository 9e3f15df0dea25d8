/* rt_native.h — C-ABI of librtnative.so, the MI355X (gfx950) implementation of the reference's compute-shader path.
 *
 * The reference has no plugin interface of its own: its host (C#) drives the kernels through Unity's
 * ComputeShader / ComputeBuffer / RenderTexture objects.  Each entry point below replaces exactly one of those
 * Unity calls as the reference makes it, with the same names for kernels, constants, buffers, textures and
 * keywords, so a host can P/Invoke this library instead of Unity's compute back end.
 * Citations are paths under BVH_ReSTIR_DL/Assets/Scripts/ unless they start with BVH_RIS_DL.
 *
 * Conventions
 *  - Every function returns 0 on success and a negative rt_status on failure; rt_last_error() gives the message
 *    (owned by the library, valid until the next call on that context).  No function throws or aborts.
 *  - Handles are opaque, owned by the library, and valid until released / the context is destroyed.
 *  - One caller thread per context.  Dispatches and uploads are enqueued in call order on one HIP stream and are
 *    asynchronous; rt_sync() and the *_get_data() calls wait for them.  Constants are snapshotted at dispatch.
 *  - A dispatch may be held back until the next call on its context shows whether the two can run as one launch
 *    (InitialTrace + TemporalReuse, and TransferToOriginal + TraceRays + AccumulateFrames, issued back to back do).  Every call that could observe the
 *    difference - another dispatch, an upload, a readback, rt_sync, the stream, mark, map and statistics calls -
 *    launches what is held first, so effects keep the order of the calls; a host that only dispatches and then waits on
 *    the stream by other means must call rt_flush (or rt_sync, or rt_stream) once after its last dispatch.
 *    rt_set_fusion(ctx, 0) switches the holding off for a context (the environment variable RT_FUSE=0 for a process).
 *  - A kernel launch that fails is reported as RT_ERR_LAUNCH with the kernel's name in rt_last_error().  Because a held
 *    dispatch is launched by a later call, its failure may be returned by that call instead (any entry point that
 *    returns a status, rt_flush and rt_sync included) - once, and never lost.
 *  - There is no CPU fallback: rt_create() fails when no gfx950 device is usable.
 */
#ifndef RT_NATIVE_H
#define RT_NATIVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx_s* rt_ctx;
typedef struct rt_shader_s* rt_shader;
typedef struct rt_buffer_s* rt_buffer;
typedef struct rt_texture_s* rt_texture;

typedef enum {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,   /* null handle, bad size, unknown shader/kernel name, ... */
    RT_ERR_NO_DEVICE = -2,     /* no usable gfx950 GPU, or a HIP runtime call failed */
    RT_ERR_OUT_OF_MEMORY = -3,
    RT_ERR_UNBOUND = -4,       /* a dispatch needs a buffer/texture/constant that was never set */
    RT_ERR_BAD_SCENE = -5,     /* scene buffers failed validation (index out of range, BVH too deep, cycle) */
    RT_ERR_LIMIT = -6,         /* a size exceeds a documented limit of an entry point */
    RT_ERR_NOT_PREPARED = -7,  /* rt_dispatch_no_alloc only: the dispatch would have to allocate device memory */
    RT_ERR_LAUNCH = -8,        /* a kernel launch failed; rt_last_error() names the kernel (may surface one call late, see above) */
    RT_ERR_BUSY = -9           /* rt_dispatch_no_alloc only: another context's dispatch still uses the GPU's shared scratch array */
} rt_status;

typedef enum {
    RT_FORMAT_R32F = 1,      /* RenderTextureFormat.RFloat (Main.cs:359) */
    RT_FORMAT_RGBA32F = 2,   /* the 3- and 4-channel render targets and input textures, kept in fp32 */
    /* 8-bit UNORM.  As an input (Texture2D RGBA32, ObjectHelper.cs:381) the texels are uploaded as 4 bytes each.  As a
     * render target (TextureHelper.cs:144-155 creates the 3- and 4-channel targets as RenderTextureFormat.Default)
     * every kernel store clamps to [0, 1] and rounds to a multiple of 1/255, as the reference's targets do; this
     * reaches the computation through TemporalReuse's hit-point movement test (ReSTIR_BVH_RayTracer.compute:613) and
     * the running average of AccumulateFrames.  Storage and readback stay fp32 texels either way. */
    RT_FORMAT_RGBA8_UNORM = 3
} rt_format;

/* ---- context ------------------------------------------------------------------------------------------- */

/* Stands in for the Unity graphics device.  device_id is the HIP device ordinal this process renders on
 * (one process per GPU).  Fails with RT_ERR_NO_DEVICE when that device is not a gfx950 part. */
int rt_create(int device_id, rt_ctx* out);
void rt_destroy(rt_ctx ctx);
/* Waits for all work enqueued on the context (Unity has no explicit equivalent; readbacks imply it). */
int rt_sync(rt_ctx ctx);
const char* rt_last_error(rt_ctx ctx);
/* The HIP stream (hipStream_t) the context enqueues on, for callers that time or order work against it.  Launches a
 * held dispatch first, like rt_flush. */
void* rt_stream(rt_ctx ctx);
/* Launches a dispatch that is being held back for merging (see the conventions above) and returns without waiting.
 * A host that keeps the stream from an earlier rt_stream() call and waits on it by its own means calls this after
 * the last dispatch of a frame.  No counterpart in the reference. */
int rt_flush(rt_ctx ctx);
/* Primary rays of the next frames beside the kernels of the frames before.  InitialTrace's primary rays (and the RIS
 * project's GenerateCandidates') depend on nothing the frames before compute; the library keeps a second copy of what
 * they overwrite while it is still being read (HitInfos, DepthBuffer, RayHitPointsA/B), makes the copy it writes the
 * current one at every such dispatch, and launches them on a stream of its own beside the kernels the dispatch stream
 * still holds; the dispatch stream waits for them before the candidate kernel.  Nothing a host can observe changes:
 * every readback, map and sync is ordered behind that wait, and buffers and textures read back what the dispatches, in
 * call order, left in them.  mode 0: off (RT_OVERLAP=0 in the environment: off for the process, whatever is set here); 1 (default): for InitialTrace dispatches that
 * cover the whole image; 2: also for dispatches on a row band (rt_set_row_band) - the rows of HitInfos, DepthBuffer
 * and RayHitPointsA/B outside the band are then undefined after the dispatch, which is fine for a host that renders
 * the same band every frame (the multi-GPU frame loop) and wrong for one that renders several bands of one frame one
 * after the other.  Device pointers from rt_buffer_map_rows / rt_texture_map_rows of those four objects are valid until
 * the next such dispatch.
 * Two frames in flight: with a second copy of Candidates and CandidatesB as well (made at the first overlapped
 * InitialTrace; 64 + 4 bytes per pixel each, the 4 being a per-copy tag that tells a readback which copy holds a pixel's
 * newest record; up to four copies and streams with rt_set_tuning "frames_in_flight"), consecutive frames share only TemporalFrameBuffer, Result / AccumulatedResult and the block tables,
 * so the library alternates the frames between two dispatch streams and orders only the launches that touch those.
 * What a host observes is unchanged; what changes is that "the dispatch stream" of rt_stream_order / rt_mark is the
 * stream of the frame being dispatched.  rt_stream() ends it for the context (the host's own work on that stream must
 * see every dispatch), rt_set_tuning "frames_in_flight" = 1 / RT_FRAMES_IN_FLIGHT=1 switch it off.  In mode 2 the
 * reservoir records of a pixel whose primary ray missed are whatever the current copy holds, and rows written through
 * rt_buffer_map_rows pointers are reported with rt_buffer_rows_written.  No counterpart in the reference. */
int rt_set_overlap(rt_ctx ctx, int mode);
/* Implementation choices that change speed only, never results; each also has an environment variable that sets the
 * default of new contexts.  Kept so that the alternatives that were measured and not adopted stay testable:
 *   "spatial_lds_max_offset"     (RT_SPATIAL_LDS, default 0)   SpatialReusePass stages the neighbourhood of each 16x16
 *                                 block in LDS when MaxOffset is at most this (0: every lane gathers for itself)
 *   "primary_background_groups"  (RT_PRIMARY_BG, default 0)    overlapped primary rays as this many resident workgroups
 *                                 that walk the block table (0: one workgroup per block)
 *   "overlap_behind_candidates"  (RT_OVERLAP_START, default 1) overlapped primary rays of the ReSTIR project start only
 *                                 after the previous frame's candidate kernel (0: as early as their inputs allow)
 *   "frames_in_flight"           (RT_FRAMES_IN_FLIGHT, default 2) 2 to 4: consecutive frames alternate between that many
 *                                 dispatch streams (see rt_set_overlap; 3 and 4 measured no faster than 2); 1: one
 *                                 dispatch stream.  Setting 2 or more after rt_stream() tells the library that the
 *                                 host no longer uses that stream
 *   "ris_frames_in_flight"       (RT_RIS_FRAMES_IN_FLIGHT, default 3) the same for the RIS project's frames, which are chained
 *                                 through Result only (a copy of Candidates - 16 bytes x candidates per pixel - per frame
 *                                 in flight): 0.664 / 0.640 / 0.622 / 0.640 ms per 1080p frame with 1 / 2 / 3 / 4
 *   "region_blocks_rays"         (default 2)  1, 2, 4 or 8: side, in 16x16-pixel blocks, of the image regions that the
 *   "region_blocks_reuse"        (default 4)  heaviest-first workgroup tables move as a whole - for the kernels that cast
 *   "region_blocks_primary"      (default 4)  secondary rays, for the reuse / copy / accumulate kernels, and for the primary
 *                                 rays.  Doubled while an image has more than 8192 regions of that size.  Setting one waits
 *                                 for the context's work and rebuilds the tables (1080p: 0.697 ms per frame with 4 / 4 / 4,
 *                                 0.686 with 2 for the secondary rays; 4K: 2.436 and 2.407 ms)
 *   "one_wave_groups"            (RT_WG64, default 15)  per-pixel kernels launched with one wave per workgroup instead
 *                                 of the four of a 16x16-pixel block: bit 0 the kernels that cast secondary rays, bit 1 the
 *                                 ReSTIR project's primary rays, bit 2 the RIS project's, bit 3 SpatialReusePass, bit 4 the
 *                                 RIS project's candidate kernel (1080p ReSTIR frame 0.675 ms with 0, 0.664 with 5, 0.683
 *                                 with 7, 0.656 with 15; 4K 2.43 / 2.35 ms with 5 / 15; RIS frame 0.587 / 0.552 / 0.569 ms
 *                                 with 0 / 5 / 21)
 *   "table_refresh_frames"       (default 8)  the regions are ranked anew every so many InitialTrace dispatches (into a
 *                                 second pair of tables, so that the frames in flight keep theirs: a refresh that made
 *                                 the frame wait for the one before cost 0.1 ms, 12 us per frame at 8)
 *   "top_nodes_lds"              (default 0) 0 to 64: the merged candidate and TraceRays launches keep the first so many
 *                                 internal BVH nodes (breadth first from the scene root into the objects' meshes) in LDS, a
 *                                 copy per wave.  Same results; 1080p frame 0.638 ms with 0, 0.651 with 8 to 32, 0.691 with 64.
 *                                 Setting it waits for the context's work; the scene's arrays are rebuilt at the next dispatch
 *   "reuse_validation"           (default 1) what a full check of the scene buffers found out about the mesh BVHs is kept
 *                                 (per context and project: 8 bytes per BVH node on the host) and relied on by the next
 *                                 check when the nodes below the scene BVH carry the same link words - a digest taken on
 *                                 the device decides - so that a reload walks only its new scene BVH (first InitialTrace
 *                                 after reloading 10 M triangles: 52 -> 4.7 ms).  0: every check walks every mesh
 *   "device_pool_mb"             (RT_DEVICE_POOL_MB, default a sixteenth of the device's memory) released device memory
 *                                 (buffers, textures, their copies, the arrays derived from a scene) is kept, up to this
 *                                 many MiB per context, for the next allocation of the same size instead of going
 *                                 through hipFree and hipMalloc: a scene reload releases and re-creates every buffer
 *                                 (80k-triangle scene at 1080p: the frame that reloads 8.5 -> 3.1 ms).  0: keep nothing
 *                                 (what is kept is freed at once).  rt_destroy frees whatever is kept
 *   "debug_stream_delay_us"      (default 0) test aid: before about every second launch its stream is held up for a
 *                                 pseudo-random time of up to this many microseconds (at most 2000), so that a missing
 *                                 order between launches of frames in flight shows as a wrong image
 *   "debug_stream_delay_mask"    (default 255) which streams that applies to: bit k = dispatch stream k, bit 7 = the
 *                                 stream of the overlapped primary rays
 * Unknown names fail with RT_ERR_INVALID_ARG.  No counterpart in the reference. */
int rt_set_tuning(rt_ctx ctx, const char* name, int value);
/* Test aid: a log of what the library puts on which of its streams.  rt_debug_trace(ctx, 1) starts a new log, (ctx, 0)
 * stops it; rt_debug_trace_dump writes it as text (at most `capacity` bytes with the terminating 0) and returns the
 * bytes the whole text needs.  One line per operation, in the order the host enqueued them:
 *   L <stream> <kernel> <object>:<r|w>[~][:<row0>-<row1>] ...   a launch and the per-pixel objects (device addresses of
 *                                               the copies) it reads / writes, with the image rows when it works on a
 *                                               band; ~ marks counts another kernel may read while they are rewritten
 *   R <stream> <event>    an event recorded          W <stream> <event>    a stream made to wait for an event
 *   H 0                   a host operation on dispatch stream 0 (a copy, a conversion, a scene rebuild: may touch anything)
 *   Y 0                   rt_sync: the host has waited for every stream
 * Streams: 0..3 the dispatch streams, 4 the overlapped primary rays' stream, 5 the transfer stream.  From the log alone
 * it can be decided whether every two launches that touch the same memory, one of them writing, are ordered by the
 * streams and events - whatever the timing (tests/harness.py does).  No counterpart in the reference. */
int rt_debug_trace(rt_ctx ctx, int on);
int64_t rt_debug_trace_dump(rt_ctx ctx, void* out, int64_t capacity);
/* For a host that moves rows itself (the multi-GPU loop's halo exchange and gather): adds to the log that work it has
 * just enqueued on `stream` (0 = the dispatch stream of the frame being dispatched, 1 = the transfer stream) reads
 * (write = 0) or writes rows [row0, row1) of the current copy of a buffer and / or a texture.  Does nothing when no
 * log is being kept. */
int rt_debug_note_rows(rt_ctx ctx, int stream, rt_buffer buf, rt_texture tex, int row0, int row1, int write);
/* enabled = 0: every dispatch is launched by the call that makes it (nothing is held or merged); 1 (the default
 * unless RT_FUSE=0 is in the environment): consecutive dispatches that can run as one launch do.  Results are
 * bit-identical either way.  No counterpart in the reference. */
int rt_set_fusion(rt_ctx ctx, int enabled);

/* Multi-GPU image sharding (no counterpart in the reference, which is single-GPU): restricts every per-pixel
 * kernel of this context to image rows [row0, row1).  Pixel coordinates, RNG seeds and buffer indexing stay
 * global, so a band computes bit for bit what the full frame computes there.  row1 <= 0 restores full frames. */
int rt_set_row_band(rt_ctx ctx, int row0, int row1);

/* ---- shaders: the ComputeShader assets referenced by Main.cs:46-48 --------------------------------------- */

/* name is one of "ReSTIR_BVH_RayTracer", "PreCalc", "PostProcessor" (the ReSTIR project's three assets) or
 * "RIS_BVH_RayTracer", "PreCalc", "PostProcesser" (the RIS project's: BVH_RIS_DL/Assets/Scripts/Main.cs:41-43,73).
 * Both projects call their pre-calculation asset "PreCalc"; which CalcTriNormals runs is decided at dispatch by
 * the buffers bound to it: "Tris" + "SceneObjects" (BVH_RIS_DL/.../ShaderHelper.cs:16,31) select the RIS
 * project's world-normal variant, "Triangles" + "Vertices" (ShaderHelper.cs:26,37) the ReSTIR project's. */
int rt_shader_open(rt_ctx ctx, const char* name, rt_shader* out);
void rt_shader_close(rt_shader sh);

/* ComputeShader.FindKernel (ComputeHelper.cs:16,23).  Returns the kernel index (>= 0) or a negative rt_status.
 * ReSTIR: InitialTrace 0, SpatialReusePass 1, TransferToOriginal 2, TemporalReuse 3, TraceRays 4
 * (ComputeShaders/ReSTIR_BVH_RayTracer.compute:1-5); RIS: GenerateCandidates 0, TraceRays 1
 * (BVH_RIS_DL/.../RIS_BVH_RayTracer.compute:1-2); PreCalc: CalcTriNormals 0; PostProcessor: AccumulateFrames 0. */
int rt_find_kernel(rt_shader sh, const char* name);

/* ComputeShader.Dispatch (ComputeHelper.cs:16,23).  Thread-group sizes are fixed per kernel as in
 * ComputeShaders/Constants.hlsl:7-9 (8x8x1 for the tracer and post-process kernels, 256x1x1 for PreCalc). */
int rt_dispatch(rt_shader sh, int kernel, int groups_x, int groups_y, int groups_z);

/* rt_dispatch for callers that cannot allocate device memory (a frame loop under HIP graph capture, a serving loop
 * or harness that forbids hipMalloc).  No counterpart in the reference.  Same arguments, same kernels and bit-identical
 * results, but the call makes no hipMalloc / hipFree or any other allocating call: what rt_dispatch allocates lazily
 * and caches - the workgroup-to-image-block table of the launch and the per-tile hit counts InitialTrace ranks it by -
 * lives instead in a fixed 1.25 MiB device array that the library holds on every GPU it runs on until the process
 * ends, and is rebuilt on the stream by every call (two small extra launches).
 *  - Limit: the launch may cover at most 4096 regions of 64x64 pixels (e.g. 4096x4096 pixels, or the rows of a row
 *    band); beyond it the call fails with RT_ERR_LIMIT and launches nothing.
 *  - What cannot be done without allocating is refused with RT_ERR_NOT_PREPARED: scene buffers whose traversal arrays
 *    are not built yet (rt_dispatch builds them at the first dispatch of a traversal kernel after a scene upload) and
 *    per-pixel buffers not yet in the library's internal layout (the first rt_dispatch that binds them converts
 *    them).  Run one frame with rt_dispatch first; rt_buffer_set_data on a reservoir or HitInfos buffer puts it
 *    back into the ABI layout (rt_buffer_get_data does not).
 *  - Launches are not timed (rt_profile_enable is ignored for them); work counters are kept as usual.
 *  - All contexts on one GPU share the array: calls that overlap in time on one GPU (different contexts, hence
 *    different streams) would race on it, so a call made while ANOTHER context's rt_dispatch_no_alloc launches are still
 *    in flight on that GPU fails with RT_ERR_BUSY and launches nothing.  Calls on one context are ordered by its stream. */
int rt_dispatch_no_alloc(rt_shader sh, int kernel, int groups_x, int groups_y, int groups_z);

/* ComputeShader.EnableKeyword / DisableKeyword (Main.cs:252-257): "VIS_REUSE", "WEIGHT_RECALC", "BRDF".
 * Unknown keywords are ignored, as Unity does. */
int rt_enable_keyword(rt_shader sh, const char* keyword);
int rt_disable_keyword(rt_shader sh, const char* keyword);

/* ComputeShader.SetInt / SetFloat / SetBool / SetInts / SetVector / SetMatrix
 * (Main.cs:138-140,190-199,217-267,395-396; ShaderHelper.cs:18-19,31,38; ObjectHelper.cs:508-511,637,647,686,696).
 * Constants live in a typeless table of 32-bit words keyed by name, exactly like a constant buffer: SetInts on
 * a float2 constant stores integer bit patterns (the reference does this for TextureAtlasDims and
 * EnvironmentMapTexDims).  Names the shader does not declare are accepted and ignored, as Unity does.
 * rt_set_vector takes 4 floats; rt_set_matrix takes 16 floats in column-major order (Unity's Matrix4x4);
 * a float3x3 constant uses the upper-left 3x3. */
int rt_set_int(rt_shader sh, const char* name, int value);
int rt_set_float(rt_shader sh, const char* name, float value);
int rt_set_bool(rt_shader sh, const char* name, int value);
int rt_set_ints(rt_shader sh, const char* name, const int* values, int count);
int rt_set_vector(rt_shader sh, const char* name, const float values[4]);
int rt_set_matrix(rt_shader sh, const char* name, const float values[16]);

/* ComputeShader.SetBuffer(kernelIndex, name, buffer) (ShaderHelper.cs:9-72). */
int rt_set_buffer(rt_shader sh, int kernel, const char* name, rt_buffer buf);
/* The RIS project's shader takes "Tris" (stride 76), "BVs", "SceneObjects", "LightObjects", "Materials" (stride 32),
 * "HitInfos" (stride 40) and "Candidates" (stride 16, float4) (BVH_RIS_DL/.../ShaderHelper.cs:7-52).
 * One buffer name has no counterpart in the reference: "TraceRecords" (stride 32, one record per pixel), bound to
 * ReSTIR kernel 0 and/or 4, or RIS kernel 0 and/or 1.  While it is bound those dispatches run instrumented variants that also write, per
 * pixel, 8 dwords: closest-hit triangle, scene object, BVH visit-order hash of the primary ray, selected light
 * object, selected light triangle, visit hash and closest-hit triangle of the shadow ray, visit hash of the rays
 * cast by TraceRays.  Parity tests compare them with the oracle's records. */
/* ComputeShader.SetTexture(kernelIndex, name, texture) (Main.cs:262-268,319-378). */
int rt_set_texture(rt_shader sh, int kernel, const char* name, rt_texture tex);

/* ---- buffers: ComputeBuffer ------------------------------------------------------------------------------ */

/* new ComputeBuffer(count, stride) (ComputeHelper.cs:75,98).  Contents start zeroed. */
int rt_buffer_create(rt_ctx ctx, int64_t count, int stride, rt_buffer* out);
/* ComputeBuffer.SetData (ComputeHelper.cs:76).  The library copies; the caller keeps `src`. */
int rt_buffer_set_data(rt_buffer buf, const void* src, size_t bytes);
/* ComputeBuffer.GetData.  Waits for enqueued work, then copies `bytes` from the start of the buffer, always in
 * the reference's array-of-structures layout, whatever layout the library keeps internally. */
int rt_buffer_get_data(rt_buffer buf, void* dst, size_t bytes);
/* ComputeBuffer.GetData(Array data, int managedBufferStartIndex, int computeBufferStartIndex, int count) — Unity's
 * partial readback: elements [first, first + count) in the reference's layout (the reference itself only reads whole
 * buffers; tests at 4K read the rows they compare). */
int rt_buffer_get_range(rt_buffer buf, void* dst, int64_t first, int64_t count);
/* ComputeBuffer.Release (ComputeHelper.cs:203-214, Main.cs:275 releases every buffer before a scene reload); NULL is
 * ignored.  Waits for enqueued work, then removes the handle from every binding of the context's shaders: a dispatch
 * that still needs it fails with RT_ERR_UNBOUND until a buffer is bound under that name again. */
void rt_buffer_release(rt_buffer buf);
int64_t rt_buffer_count(rt_buffer buf);
int rt_buffer_stride(rt_buffer buf);

/* ---- textures: RenderTexture / Texture2D ----------------------------------------------------------------- */

/* TextureHelper.CreateTexture (TextureHelper.cs:118-156) and `new RenderTexture(..., RFloat)` (Main.cs:359).
 * Contents start zeroed.  Reads outside the texture return 0 and writes outside are dropped. */
int rt_texture_create(rt_ctx ctx, int width, int height, rt_format format, rt_texture* out);
/* Texture2D pixel upload; `src` is width*height texels of the texture's format, row-major, row 0 first. */
int rt_texture_set_data(rt_texture tex, const void* src, size_t bytes);
/* Readback (RenderTexture -> Texture2D.ReadPixels, Main.cs:544-557): fp32 texels (1 or 4 floats per texel). */
int rt_texture_get_data(rt_texture tex, void* dst, size_t bytes);
/* Rows [row0, row1) only (Texture2D.ReadPixels with a source rectangle), fp32 texels. */
int rt_texture_get_rows(rt_texture tex, void* dst, int row0, int row1);
/* RenderTexture.Release (Main.cs:512); NULL is ignored.  Unbinds the handle like rt_buffer_release. */
void rt_texture_release(rt_texture tex);

/* ---- multi-GPU row access (no counterpart in the reference) ----------------------------------------------- */

/* Device memory holding image rows [row0, row1) of a per-pixel buffer (`width` records per row) or of a texture,
 * in the layout the library currently keeps it in: a 52-byte-record buffer is one contiguous run (the ABI layout, or
 * the padded 64-byte records the reservoirs gathered by SpatialReusePass are kept in) or four runs (plane layout).  A multi-GPU host maps the border rows of the reservoir buffers and of
 * DepthBuffer right before each exchange and sends/receives them between the per-GPU processes (RCCL); row r of
 * one process's buffer corresponds to row r of every other's, because each process keeps full-frame buffers and
 * only renders its band.  The pointers are valid until the next call that uploads to or releases the object (for
 * HitInfos, DepthBuffer and RayHitPointsA/B also: until the next InitialTrace / GenerateCandidates dispatch, see
 * rt_set_overlap). */
typedef struct {
    void* ptr;
    size_t bytes;
} rt_segment;
int rt_buffer_map_rows(rt_buffer buf, int width, int row0, int row1, rt_segment out[4], int* count);
/* After the host (or a transfer) has written rows [row0, row1) through pointers from rt_buffer_map_rows: makes those
 * rows of the current copy the ones a readback returns.  Needed only in overlap mode 2 for a reservoir buffer the
 * library keeps two copies of (in the other modes rt_buffer_map_rows has already seen to it); harmless otherwise. */
int rt_buffer_rows_written(rt_buffer buf, int width, int row0, int row1);
int rt_texture_map_rows(rt_texture tex, int row0, int row1, rt_segment* out);

/* ---- second stream for transfers (no counterpart in the reference) ---------------------------------------- */

/* A second HIP stream (hipStream_t) of the context, for transfers that should overlap the kernels: a multi-GPU host
 * puts its halo exchanges and frame-buffer gathers on it.  Created at the first call; rt_sync waits for both streams. */
void* rt_transfer_stream(rt_ctx ctx);
/* Orders the two streams against each other: everything enqueued so far on stream `first` (0 = the dispatch stream,
 * 1 = the transfer stream) completes before anything enqueued LATER on the other stream starts.  Nothing blocks on the
 * host. */
int rt_stream_order(rt_ctx ctx, int first);
/* Time stamps for measuring that overlap: rt_mark records stamp `slot` (0..63) at the current end of a stream
 * (0 or 1); rt_mark_elapsed waits for both stamps and returns the milliseconds from stamp a to stamp b. */
int rt_mark(rt_ctx ctx, int stream, int slot);
int rt_mark_elapsed(rt_ctx ctx, int slot_a, int slot_b, double* milliseconds);
/* A stamp as an ordering point: whatever is enqueued LATER on `stream` (0 = the dispatch stream of the frame being
 * dispatched, 1 = the transfer stream) starts only when the work that preceded the last rt_mark of `slot` is done.
 * Unlike rt_stream_order the point waited for need not be the present end of the other stream: a multi-GPU host
 * stamps the end of frame f, goes on enqueueing frame f+1, and later lets the gather of frame f wait for the stamp.
 * Fails with RT_ERR_INVALID_ARG for a slot that was never stamped.  Nothing blocks on the host. */
int rt_wait_mark(rt_ctx ctx, int stream, int slot);

/* ---- measurement (no counterpart in the reference) -------------------------------------------------------- */

/* Work counters of a (shader, kernel) pair, summed over every dispatch since they were last read.
 * rt_enable_stats level: 0 off (default), 1 counts rays and active pixels (negligible cost), 2 counts everything
 * (runs instrumented kernel variants).  rt_get_stats waits for enqueued work, returns the sums and resets them. */
typedef struct {
    uint64_t rays;       /* GetRayCollision calls */
    uint64_t pops;       /* BVH nodes popped */
    uint64_t bvChecks;   /* child boxes tested */
    uint64_t objVisits;  /* scene-object leaf iterations */
    uint64_t triIters;   /* leaf triangle-loop iterations */
    uint64_t triChecks;  /* ray/triangle tests (triangles passing the back-face pre-cull) */
    uint64_t validNeighbours; /* SpatialReusePass neighbour reservoirs fetched */
    uint64_t activePixels;
} rt_dispatch_stats;
int rt_enable_stats(rt_ctx ctx, int level);
int rt_get_stats(rt_shader sh, int kernel, rt_dispatch_stats* out);

/* Per-kernel profile.  One dispatch may run several GPU kernels; they are named "primary", "candidates" (the two
 * halves of InitialTrace), "temporal", "spatial", "transfer", "trace_rays", "accumulate", "calc_tri_normals".
 * With timing enabled every launch is bracketed by HIP events on the context's stream.  rt_profile_get waits for
 * enqueued work and returns launches, summed milliseconds and summed counters since the last rt_profile_reset
 * (it does not reset; rt_get_stats and rt_profile_reset do). */
typedef struct {
    uint64_t launches;
    double milliseconds;
    rt_dispatch_stats stats;
} rt_kernel_profile;
int rt_profile_enable(rt_ctx ctx, int timing);
/* Restricts the event brackets to the launches of one named kernel (NULL or "" = every kernel again).  A pair of
 * events costs a few microseconds of stream time per launch, which matters when a whole frame is being timed. */
int rt_profile_select(rt_ctx ctx, const char* kernel_name);
int rt_profile_reset(rt_ctx ctx);
int rt_profile_get(rt_ctx ctx, const char* kernel_name, rt_kernel_profile* out);

#ifdef __cplusplus
}
#endif
#endif /* RT_NATIVE_H */
