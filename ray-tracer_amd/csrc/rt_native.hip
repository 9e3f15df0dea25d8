// rt_native.hip — host side of librtnative.so: the C-ABI of include/rt_native.h on top of the HIP runtime.
//
// Mirrors the slice of Unity's ComputeShader / ComputeBuffer / RenderTexture API the reference's host uses
// (BVH_ReSTIR_DL/Assets/Scripts/ComputeHelper.cs, ShaderHelper.cs, Main.cs): named constants in a typeless
// word table, named buffer/texture bindings per kernel index, keyword toggles, dispatch by kernel index.
// Everything runs on one HIP stream per context.  There is no CPU fallback anywhere in this file.
#include "rt_kernels.hip"
#include "rt_kernels_ris.hip"

#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <mutex>
#include <string>
#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/rt_native.h"

using namespace rtn;

namespace {

enum ShaderKind { SH_RESTIR = 0, SH_RIS = 1, SH_PRECALC = 2, SH_POSTPROC = 3 };
enum BufLayout { LAYOUT_AOS = 0, LAYOUT_RESERVOIR_RECS = 1, LAYOUT_HIT_PLANES = 2, LAYOUT_RESERVOIR_PLANES = 3 };
// the kernels a dispatch is made of (one ABI dispatch may launch several)
enum InternalKernel { IK_PRIMARY = 0, IK_CANDIDATES, IK_TEMPORAL, IK_SPATIAL, IK_TRANSFER, IK_TRACE_RAYS, IK_ACCUMULATE, IK_CALC_NORMALS,
                      IK_RIS_PRIMARY, IK_RIS_CANDIDATES, IK_RIS_TRACE_RAYS, IK_RIS_CALC_NORMALS, IK_COUNT };
const char* const IK_NAMES[IK_COUNT] = {"primary", "candidates", "temporal", "spatial", "transfer", "trace_rays", "accumulate", "calc_tri_normals",
                                        "ris_primary", "ris_candidates", "ris_trace_rays", "ris_calc_tri_normals"};

}  // namespace

// Scratch of rt_dispatch_no_alloc: the workgroup -> block table of one launch (TABLE words) followed by the per-wave
// hit counts the table is ranked by (COST words).  File scope, so it exists once per GPU for the life of the process;
// `scratch` itself is only a host-side handle: the device address is taken once per device with hipGetSymbolAddress
// (rt_create) and every use below goes through that pointer.
constexpr unsigned NOALLOC_MAX_REGIONS = 4096;                                                 // 64x64-pixel regions per launch
static_assert(NOALLOC_MAX_REGIONS <= (unsigned)ORDER_MAX_REGIONS, "the ordering kernel must hold the scratch table's regions");
constexpr size_t NOALLOC_TABLE_WORDS = (size_t)NOALLOC_MAX_REGIONS * RT_REGION * RT_REGION;    // one word per 16x16-pixel block
constexpr size_t NOALLOC_COST_WORDS = NOALLOC_TABLE_WORDS * 4;                                 // one word per 8x8-pixel tile
static __device__ __align__(256) uint32_t scratch[NOALLOC_TABLE_WORDS + NOALLOC_COST_WORDS];   // 1.25 MiB

struct rt_ctx_s;
// The context whose rt_dispatch_no_alloc last used each GPU's scratch array (guarded by scratchMutex): a call from
// another context is refused while that one's launches are still in flight.
constexpr int SCRATCH_MAX_DEVICES = 64;
static std::mutex scratchMutex;
static rt_ctx_s* scratchOwner[SCRATCH_MAX_DEVICES] = {};

constexpr int MAX_FRAMES = 4;   // frames in flight at most: dispatch streams, copies of a per-pixel object
// The further copies of a per-pixel object the frames alternate between, the one least recently current first.
struct Copies {
    void* alt[MAX_FRAMES - 1] = {};
    uint32_t* tag[MAX_FRAMES - 1] = {};   // record buffers only: the copies' tags (see Recs)
    int n = 0;
};

struct Link { int32_t indexA, indexB; };   // the two index words of a BV node

struct rt_ctx_s {
    int device = 0;
    // The dispatch stream.  Every use goes through S(ctx) or SK(ctx), which first launch a held-back dispatch (see Deferred).
    hipStream_t stream0 = nullptr;
    // Primary rays of the NEXT frames beside the kernels of the frames before.  InitialTrace's primary-ray kernel (and the
    // RIS project's) reads nothing any kernel of the frame sequence produces; it only OVERWRITES per-pixel data that
    // kernels of the frame before still read: HitInfos and DepthBuffer (candidate kernel, SpatialReusePass,
    // TransferToOriginal, TraceRays) and the hit-point target of its parity (TemporalReuse reads both targets).  The
    // library keeps a second copy of each of those, writes the copy that is NOT current and makes it the current one
    // at the dispatch, and launches the primary rays on `stream1`, where they queue behind one another and run beside
    // whatever the frame streams hold; the frame's stream waits for them (`evPrimary`) before the candidate kernel.
    // Everything a host can observe stays in call order: every readback, map and sync goes through stream0 behind all
    // other streams, and reads the current copies.
    // `mainDirty`: something other than the frames' own kernels was put on stream0 since (a scene rebuild, a cleared new
    // buffer, primary rays that could not be overlapped, work of the host's own after rt_stream): the next primary
    // rays then wait for all of it.
    // overlap: 0 off, 1 (default) InitialTrace dispatches that cover the whole image, 2 also those on a row band (the
    // rows outside the band are then undefined in the objects with two copies: a multi-GPU host never reads them).
    hipStream_t stream1 = nullptr;
    hipEvent_t evPrimary = nullptr;
    // Two frames in flight.  With a second copy of Candidates and CandidatesB as well, the kernels of frame f+1 touch
    // nothing the kernels of frame f touch, except TemporalFrameBuffer (TemporalReuse of f+1 reads what that of f wrote),
    // Result / AccumulatedResult (TraceRays and AccumulateFrames of f+1 after those of f) and the block tables.  So the
    // frames alternate between two dispatch streams, frameStream[0] (= stream0) and frameStream[1]: frame f+1's candidate
    // kernel and reuse passes run beside frame f's reuse passes and TraceRays, which is what fills the GPU when a launch
    // covers a row band only (a 136-row band of a 3840x2160 frame: 0.48 -> 0.37 ms per frame).  Orders that are kept:
    //  * a launch that touches one of the shared objects waits for the last launch on the OTHER frame stream that did
    //    (evShared[object][stream]); launches on one stream are in order anyway;
    //  * the copies a frame writes were last current a few changes of copies ago: at every InitialTrace that changes
    //    copies, the ends of all frame streams are recorded (evBefore), and the streams that will write the copies that
    //    become current (stream1 and the new frame stream) wait for the record made when those copies were left;
    //  * a frame kernel that is given a record buffer the frame's InitialTrace did not change copies of (rt_buffer_s::epoch)
    //    first waits for the other frame stream (joinOthers);
    //  * anything else - an upload, a readback, a scene rebuild, rt_stream - goes to stream0 behind BOTH frame streams
    //    (S(): collapse), and frame launches that follow on another frame stream wait for stream0 (dirty);
    //  * the table of the frame streams' kernels (BlockMap::dyn) is rewritten only with both frame streams joined.
    // The same with three or four frames in flight (framesInFlight streams, as many copies): what a row band of a
    // multi-GPU run gains from (a band's launches are far from filling the chip).
    // `cur`: the frame stream of the frame being dispatched; it moves on at every InitialTrace whose primary rays are
    // overlapped, when framesInFlight is above 1 (rt_set_tuning "frames_in_flight", RT_FRAMES_IN_FLIGHT; 1 = one frame stream)
    // and the host has not asked for the dispatch stream itself (rt_stream: streamShared, one frame stream from then on).
    hipStream_t frameStream[MAX_FRAMES] = {};
    int cur = 0;
    int framesInFlight = 2;        // 1..MAX_FRAMES dispatch streams the ReSTIR project's frames alternate between
    int risFramesInFlight = 3;     // the same for the RIS project's frames (chained through Result only: 0.664 / 0.640 / 0.622 / 0.640 ms
                                   // per 1080p frame with 1 / 2 / 3 / 4)
    bool streamShared = false;
    bool ahead[MAX_FRAMES] = {};     // [s >= 1] frameStream[s] holds work stream0 has not been made to wait for
    bool dirty[MAX_FRAMES] = {};     // [s >= 1] stream0 holds work frameStream[s] has not been made to wait for
    hipEvent_t evJoin[MAX_FRAMES] = {}, evOther[MAX_FRAMES] = {}, evTail0 = nullptr, evTable = nullptr;
    hipEvent_t evTableAside = nullptr;   // behind a table refresh on stream1 (refreshTables)
    bool asidePending = false;           // ... that no frame stream has been made to wait for yet
    enum { SH_TEMPORAL = 0, SH_RESULT = 1, SH_COUNT = 2 };
    hipEvent_t evShared[SH_COUNT][MAX_FRAMES] = {};
    bool sharedValid[SH_COUNT][MAX_FRAMES] = {};
    // evBefore[n % BEFORE_RING][s]: where frame stream s stood when the n-th change of copies was dispatched, i.e.
    // everything dispatched before it.  An object with C copies writes, at change n, the copy that was current until
    // change n - C + 1: its last users precede that change.
    static constexpr int BEFORE_RING = MAX_FRAMES + 1;
    hipEvent_t evBefore[BEFORE_RING][MAX_FRAMES] = {};
    bool beforeValid[BEFORE_RING][MAX_FRAMES] = {};
    uint64_t changes = 0;          // changes of copies so far
    uint64_t epoch = 2;            // number of the frame being dispatched, for the record tags (+1 whenever copies change)
    std::vector<rt_buffer> tagged; // the record buffers that have two copies
    // By choice (the default) the primary rays also wait for `evAfterCandidates`, recorded on stream0 behind every
    // candidate-kernel launch: they then run beside the previous frame's SpatialReusePass and TraceRays launches only,
    // not beside its candidate kernel, which measured a little faster on one GPU (0.745 against 0.756 ms per 1080p frame,
    // 2.43 against 2.47 ms at 3840x2160).
    hipEvent_t evAfterCandidates = nullptr;
    hipEvent_t afterCandidatesEv = nullptr;   // the event that marks that point now: evAfterCandidates, or one recorded there anyway
    bool afterCandidatesValid = false;
    bool behindCandidates = true;
    bool mainDirty = true;
    int overlap = 1;
    // Overlapped primary rays as a "background" launch (k_primary<.., BG>): bgGroups resident workgroups (0 = an
    // ordinary launch) walk the block table through bgCounter (8 words, one per XCD, cleared on stream1 before each launch).
    int bgGroups = 0;
    uint32_t* bgCounter = nullptr;
    // Test aid (rt_set_tuning "debug_stream_delay_us"): before about every second launch the stream it goes to is held up
    // for a pseudo-random time up to this many microseconds, which moves the kernels of the frames in flight against
    // one another in ways a quiet machine never shows: a missing wait between two launches then changes the image.
    // "debug_stream_delay_mask": which streams are held up - bit k: dispatch stream k, bit 7: the primary rays' stream
    // (default all).  Holding up ONE stream makes the others run ahead of it wherever nothing orders them.
    // Test aid (rt_debug_trace): a log of what was put on which stream - every launch with the per-pixel objects it reads
    // and writes, every event record and wait, every host operation on stream0 - from which a checker derives whether
    // any two launches that touch the same memory, at least one of them writing, are ordered by the streams and events
    // alone (tests/harness.py: launch_order_violations).  Timing plays no part in that.
    struct TraceAccess { const void* obj; bool write; bool relaxed; int row0, row1; };   // image rows [row0, row1); row1 < 0: all of it
    struct TraceOp { char kind; int stream; const void* event; const char* name; int nAcc; TraceAccess acc[16]; };
    bool tracing = false;
    std::vector<TraceOp> traceOps;
    int debugDelayUs = 0;
    int debugDelayMask = 0xff;
    uint32_t debugDelayState = 12345u;
    int spatialLdsMaxOffset = 0;  // largest MaxOffset for which SpatialReusePass stages its neighbourhood in LDS (0 = never)
    bool bigLdsAllowed = false;   // k_spatial_lds may ask for more than 64 KiB of dynamic LDS on this device
    hipStream_t transferStream = nullptr;  // created by rt_transfer_stream
    hipEvent_t orderEvent[2] = {nullptr, nullptr};
    hipEvent_t marks[64] = {};
    std::string err;
    int row0 = 0, row1 = 0;
    int statsLevel = 0;  // 0 off, 1 rays + active pixels, 2 all counters (instrumented kernel variants)
    unsigned long long* statsDev = nullptr;  // IK_COUNT x STAT_SHARDS x 8 counters, accumulated until read
    // workgroup -> image block table of the per-pixel kernels (see buildBlockMap)
    // dev: the static order; dyn: the heaviest-first order k_order_blocks rebuilds after InitialTrace's primary rays, from the
    // per-wave hit counts in cost until a TraceRays has run on this grid, then from the steps its waves needed (steps);
    // null when the grid has more regions than that kernel sorts
    struct BlockMap { int bx, by; uint32_t* dev; unsigned n; uint32_t* dyn; uint32_t* cost; unsigned nDyn; bool dynValid; unsigned age; uint32_t* steps; bool stepsValid; bool rankedBySteps; uint32_t* dynHits; bool pendingDyn; bool pendingMeasured;
                      // region sizes (blocks per side) of the three heaviest-first tables, and the tables that differ from `dyn`:
                      // dynHits (ranked by hit pixels: the primary rays) and dynRays (ranked by steps like dyn: the kernels that
                      // cast secondary rays; the same pointer as dyn when the region sizes are equal)
                      int rs, rsHits, rsRays; unsigned nDynHits; uint32_t* dynRays; unsigned nDynRays;
                      // second buffers for dyn and dynRays: a refresh writes the pair that is not in use and then swaps, so that it
                      // need not wait for the kernels of the frames in flight that still walk the old tables; swapChange = the
                      // context's count of copy changes when the spare pair was last in use
                      uint32_t* dynSpare; uint32_t* dynRaysSpare; uint64_t swapChange;
                      bool pendingSwap; };   // the spare pair holds tables built aside that the next InitialTrace takes up
    std::vector<BlockMap> blockMaps;  // one per band height in use (a multi-GPU host alternates between a few)
    // Blocks per side of a region in the heaviest-first tables of [0] the reuse, copy and accumulate kernels, [1] the
    // primary rays, [2] the kernels that cast secondary rays (tuning "region_blocks_reuse" / "_primary" / "_rays").
    // Large regions keep an XCD's L2 on one part of the image and of the BVH; small ones let the heaviest-first order
    // balance the XCDs more finely, which is what the secondary rays (few, long, uneven) gain from.
    int regionBlocks[3] = {RT_REGION, RT_REGION, 2};
    // Per-pixel kernels launched with one wave per workgroup (tuning "one_wave_groups", RT_WG64): bit 0 the kernels that
    // cast secondary rays, bit 1 the ReSTIR project's primary rays, bit 2 the RIS project's, bit 3 SpatialReusePass,
    // bit 4 the RIS project's candidate kernel.  A wave's slot and LDS are free for the next workgroup as soon as that
    // wave ends - what counts when kernels of several frames share the chip; and kernels that run beside one another
    // should ask for slots in the same unit (primary rays in single waves beside reuse passes in fours starve the passes).
    int oneWaveGroups = [] { const char* e = getenv("RT_WG64"); return e ? atoi(e) & 31 : 15; }();
    int tableRefreshFrames = 8;   // the tables are ranked anew every so many InitialTrace dispatches (tuning "table_refresh_frames")
    uint32_t* scratch = nullptr;  // this device's address of the file-scope `scratch` array
    bool noAlloc = false;         // set for the duration of rt_dispatch_no_alloc
    BlockMap noAllocMap = {};     // the table of the launch in flight when noAlloc is set (lives in `scratch`)
    // per-internal-kernel timing with HIP events on the context's stream
    bool timing = false;
    int timingOnly = -1;  // IK_* of the only kernel whose launches are bracketed by events, -1 = all
    struct Timed { int kernel; hipEvent_t start, stop; };
    std::vector<Timed> pending;
    std::vector<hipEvent_t> freeEvents;
    double timeMs[IK_COUNT] = {};
    uint64_t launches[IK_COUNT] = {};
    uint64_t versionCounter = 1;
    // A dispatch that is held back until the next call shows whether it can be merged with what follows it.
    // TransferToOriginal, TraceRays and AccumulateFrames each touch only the thread's own pixel, and the reference issues
    // them back to back (Main.cs:387-420): TransferToOriginal is held, a TraceRays over the same buffers, rows and grid
    // takes its copy along (the reservoir it copies is the one TraceRays reads) and is held in turn, and an
    // AccumulateFrames over the same Result adds its running average to the same launch.  One launch instead of three:
    // one tail instead of three, and the copied reservoir and the stored colour are not read back from memory.
    // Any other call - another dispatch, an upload, a readback, a sync, a stream or timing query - first launches what
    // is held, unmerged, so the order of effects is the order of the calls.  Constants are snapshotted at dispatch as
    // always.  Not held: instrumented variants, rt_dispatch_no_alloc, RT_FUSE=0.
    struct Deferred {
        int kind = 0;               // 0 nothing, 1 TransferToOriginal, 2 TraceRays (withTransfer: it carries the copy),
                                    // 3 InitialTrace's candidate kernel, 4 the RIS project's TraceRays
        KParams P;
        unsigned blocks = 0;
        size_t ldsBytes = 0;
        bool withTransfer = false;
        bool temporalWaited = false;   // kind 3: the stream already waits for the other frames' TemporalFrameBuffer writes
        BlockMap* grid = nullptr;
    } held;
    bool fuse = true;
    // Status of a launch whose dispatch call had already returned (a held dispatch is launched by a later call), kept
    // until the next int-returning entry point hands it to the caller (see finish()).
    int stickyRc = 0;
    std::string stickyMsg;
    std::vector<rt_shader> shaders;   // the open shaders of this context: releasing a buffer or texture unbinds it from them
    hipEvent_t noAllocDone = nullptr; // recorded after every rt_dispatch_no_alloc: tells when this context has left the shared scratch
    // prepared traversal data (derived from the bound scene buffers): a small cache, least recently used entry replaced,
    // so that several shaders with different scene buffers on one context do not rebuild each other's arrays
    struct Prep {
        float4* wide = nullptr; size_t wideBytes = 0;
        float4* btri = nullptr; size_t btriBytes = 0;
        uint32_t* objRoot = nullptr; size_t objRootBytes = 0;
        float4* top = nullptr; size_t topBytes = 0;              // table of top nodes (ENTRY_TOP), when asked for
        uint32_t* objRootTop = nullptr; size_t objRootTopBytes = 0;
        int topCount = 0;
        uint32_t sceneRootEntryTop = 0;
        uint64_t vBVs = 0, vTris = 0, vVerts = 0, vObjs = 0;
        const void *pBVs = nullptr, *pTris = nullptr, *pVerts = nullptr, *pObjs = nullptr;
        uint32_t sceneStart = 0xffffffffu;
        uint32_t sceneRootEntry = 0;
        int stackDepth = 0;
        bool ris = false;
        bool valid = false;
        uint64_t lastUse = 0;
    };
    bool launchingHeld = false;   // inside launchHeld (the launches that may use the table of top nodes): for the trace
    // What the last full validation of a scene found out about the part of BVs BELOW its scene BVH (the mesh BVHs): a
    // reload uploads that part again unchanged (ObjectHelper.ConstructScene keeps the meshes and appends or replaces
    // the scene BVH), and a digest of its link words, taken on the device, is enough to know.  One per project.
    struct MeshInfo { int depth; uint32_t maxNode; int64_t leafEnd; };
    struct Validated {
        bool valid = false;
        size_t nPrefix = 0;                        // nodes [0, nPrefix) lie below every scene-level node of the validated scene
        unsigned long long digest[2] = {0, 0};     // of their link words
        std::vector<Link> links;                   // host copy of the link words of the BVs last prepared (its first nPrefix are what counts)
        std::map<uint32_t, MeshInfo> meshes;       // the mesh BVHs validated together, by root: depth, highest node, end of the highest leaf range
    } validated[2];
    bool reuseValidation = true;   // tuning reuse_validation
    int topNodes = 0;   // tuning top_nodes_lds: slots of the table of top nodes the merged ray kernels keep in LDS (0: none)
    static constexpr size_t MAX_PREPS = 4;
    std::vector<Prep> preps;
    uint64_t prepClock = 0;
    // Device memory the context has let go of, kept for the next allocation of the same size (see devAlloc).
    struct DevPool {
        struct Idle { void* p; uint64_t stamp; };
        std::multimap<size_t, Idle> idle;             // by size class
        std::unordered_map<void*, size_t> sizeOf;     // every block devAlloc has handed out and not yet taken back
        size_t idleBytes = 0, capBytes = 0;
        uint64_t clock = 0;
    } pool;
};

struct rt_buffer_s {
    rt_ctx ctx;
    int64_t count;
    int stride;
    size_t bytes;
    void* dev;
    int layout;
    uint64_t version;
    Copies more;          // HitInfos, Candidates, CandidatesB: the copies the next frames work on (see rt_ctx_s::stream1, frameStream)
    // Candidates / CandidatesB only (frames write them for hit pixels alone): per copy and pixel, the number of the
    // frame that last wrote the pixel's record into that copy (see Recs).  A readback takes each pixel from the copy
    // with the larger tag.
    uint32_t* tag = nullptr;     // of the copy `dev` (those of the others: more.tag)
    int tagGroup = 4;            // float4s per tag: a 64-byte reservoir record, or the candidates of one pixel ray (RIS)
    uint64_t epoch = 0;          // rt_ctx_s::epoch when the copies last changed places
    bool zeroed = false;         // still as rt_buffer_create cleared it: nothing has been uploaded, dispatched on or mapped
};

struct rt_texture_s {
    rt_ctx ctx;
    int w, h;
    rt_format format;
    void* dev;     // float (R32F) or float4 (others) texels
    size_t bytes;  // device bytes
    uint64_t version;  // bumped by every upload
    Copies more;          // DepthBuffer, RayHitPointsA/B: the copies the next frames' primary rays write (see rt_ctx_s::stream1)
};

struct rt_shader_s {
    rt_ctx ctx;
    int kind;
    std::map<std::string, std::vector<uint32_t>> constants;
    std::map<std::string, rt_buffer> buffers[8];
    std::map<std::string, rt_texture> textures[8];
    bool kwVisReuse = false, kwWeightRecalc = false, kwBrdf = false;
};

namespace {

int fail(rt_ctx ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}
#define HIP_TRY(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_NO_DEVICE,   \
                        "%s failed: %s", #call, hipGetErrorString(e_));                             \
    } while (0)

int useDevice(rt_ctx ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return RT_OK;
}

// Device allocations that come and go with the host's buffers.  A scene reload (Main.SetData) releases every buffer
// and creates buffers of the same sizes again; hipFree and hipMalloc cost 130 and 60 us a call, about thirty calls of
// each per reload, which is most of what a reload of a small scene takes.  So a released block is kept (up to
// pool.capBytes in all, the blocks released longest ago making room) and handed to the next request of its size
// class.  Sizes are rounded up to sixteen classes per power of two (at most 6 % more than asked for): the ReSTIR
// project's BVs buffer grows by a few nodes with every reload (its builder appends the new scene BVH), and a block
// that never fits again would only fill the list.  A block may be handed out while work that used it is still
// enqueued: whoever takes a block first writes it on stream0 behind everything the context has enqueued (S(ctx): the
// clearing of a new buffer, a conversion, the scene's derived arrays), so the order of effects is the streams' order,
// as with memory fresh from hipMalloc.  hipMalloc running out of memory empties the list and is tried again.
size_t devSizeClass(size_t bytes) {
    if (bytes <= 4096) return 4096;
    int top = 63 - __builtin_clzll((unsigned long long)bytes);   // 2^top <= bytes
    const size_t step = (size_t)1 << (top - 4);
    return (bytes + step - 1) & ~(step - 1);
}
void devPoolTrim(rt_ctx ctx, size_t keepBytes) {   // the blocks released longest ago go first
    auto& pool = ctx->pool;
    while (pool.idleBytes > keepBytes && !pool.idle.empty()) {
        auto oldest = pool.idle.begin();
        for (auto it = pool.idle.begin(); it != pool.idle.end(); ++it) if (it->second.stamp < oldest->second.stamp) oldest = it;
        hipError_t e = hipFree(oldest->second.p);
        (void)e;
        pool.idleBytes -= oldest->first;
        pool.idle.erase(oldest);
    }
}
hipError_t devAlloc(rt_ctx ctx, void** p, size_t bytes) {
    auto& pool = ctx->pool;
    *p = nullptr;
    const size_t size = devSizeClass(bytes);
    auto it = pool.idle.find(size);
    if (it != pool.idle.end()) {
        *p = it->second.p;
        pool.idle.erase(it);
        pool.idleBytes -= size;
        pool.sizeOf[*p] = size;
        return hipSuccess;
    }
    hipError_t e = hipMalloc(p, size);
    if (e == hipErrorOutOfMemory && !pool.idle.empty()) {
        e = hipGetLastError();   // (clears it)
        devPoolTrim(ctx, 0);
        e = hipMalloc(p, size);
    }
    if (e == hipSuccess) pool.sizeOf[*p] = size;
    else *p = nullptr;
    return e;
}
hipError_t devFree(rt_ctx ctx, void* p) {
    if (!p) return hipSuccess;
    auto& pool = ctx->pool;
    auto it = pool.sizeOf.find(p);
    if (it == pool.sizeOf.end()) return hipFree(p);
    const size_t size = it->second;
    pool.sizeOf.erase(it);
    if (size > pool.capBytes) return hipFree(p);
    devPoolTrim(ctx, pool.capBytes - size);
    pool.idle.emplace(size, rt_ctx_s::DevPool::Idle{p, ++pool.clock});
    pool.idleBytes += size;
    return hipSuccess;
}
int traceStreamIndex(rt_ctx ctx, hipStream_t s) {
    for (int k = 0; k < MAX_FRAMES; k++) if (s == ctx->frameStream[k] && s) return k;
    if (s == ctx->stream0) return 0;
    if (s == ctx->stream1) return MAX_FRAMES;
    if (s == ctx->transferStream) return MAX_FRAMES + 1;
    return MAX_FRAMES + 2;
}
// Every event of the library that orders streams goes through these two (the timing events of LaunchScope do not).
hipError_t evRecord(rt_ctx ctx, hipEvent_t ev, hipStream_t stream) {
    if (ctx->tracing) ctx->traceOps.push_back({'R', traceStreamIndex(ctx, stream), (const void*)ev, "", 0, {}});
    return hipEventRecord(ev, stream);
}
hipError_t evWait(rt_ctx ctx, hipStream_t stream, hipEvent_t ev) {
    if (ctx->tracing) ctx->traceOps.push_back({'W', traceStreamIndex(ctx, stream), (const void*)ev, "", 0, {}});
    return hipStreamWaitEvent(stream, ev, 0);
}
struct TraceLaunch {   // collects the accesses of one launch
    rt_ctx_s::TraceOp op;
    TraceLaunch(rt_ctx ctx, hipStream_t stream, const char* name) { op = {'L', traceStreamIndex(ctx, stream), nullptr, name, 0, {}}; }
    int row0 = 0, row1 = -1;   // the rows the launch works on (a row band), for the accesses added with r() and w()
    void add(const void* obj, bool write, bool relaxed = false, int r0 = 0, int r1 = -1) {
        if (!obj || op.nAcc >= 16) return;
        op.acc[op.nAcc++] = {obj, write, relaxed, r0, r1};
    }
    void r(const void* obj) { add(obj, false, false, row0, row1); }
    void w(const void* obj) { add(obj, true, false, row0, row1); }
    void rAround(const void* obj, int reach) { add(obj, false, false, row1 < 0 ? 0 : (row0 - reach > 0 ? row0 - reach : 0), row1 < 0 ? -1 : row1 + reach); }
};
// What a launch reads and writes, for the trace (rt_ctx_s::TraceOp).  `merged`: the launch also does the work of the
// dispatch merged into it (TemporalReuse into the candidate kernel, TransferToOriginal into TraceRays); `acc`: and
// AccumulateFrames.  The per-wave cost and step counts are relaxed accesses: k_order_blocks may read them while they are
// being rewritten (it reads every count once, so its table is a complete one whatever it reads).
void traceLaunch(rt_ctx ctx, hipStream_t stream, int ik, const KParams* Pp, bool merged, bool acc) {
    if (!ctx->tracing) return;
    TraceLaunch t(ctx, stream, IK_NAMES[ik]);
    if (Pp) {
        const KParams& P = *Pp;
        t.add(P.blockMap, false);
        if (P.row1 > P.row0) { t.row0 = P.row0; t.row1 = P.row1; }
        switch (ik) {
            case IK_PRIMARY:
                t.w(P.hit.p0); t.w(P.depth.r); t.w(P.normals.rgba); t.w(P.debugOverlay.rgba);
                t.w(P.FrameCount % 2 == 0 ? P.hitA.rgba : P.hitB.rgba);
                t.add(P.blockCost, true, true);
                break;
            case IK_CANDIDATES:
                t.r(P.hit.p0); t.w(P.cand.q);
                if (merged) { t.r(P.depth.r); t.w(P.temporal.p0); t.r(P.hitA.rgba); t.r(P.hitB.rgba); }
                if (ctx->launchingHeld && P.topCount > 0) t.r(P.top);   // (the variant that keeps the top nodes in LDS)
                break;
            case IK_TEMPORAL:
                t.w(P.cand.q); t.w(P.temporal.p0); t.r(P.depth.r); t.r(P.hitA.rgba); t.r(P.hitB.rgba);
                break;
            case IK_SPATIAL:
                // (neighbours up to MaxOffset rows away, and one more row up: a negative x wraps into the row before)
                t.rAround(P.ReuseBufferCycle ? P.cand.q : P.candB.q, P.MaxOffset + 1);
                t.w(P.ReuseBufferCycle ? P.candB.q : P.cand.q);
                t.rAround(P.depth.r, P.MaxOffset + 1);
                break;
            case IK_TRANSFER:
                t.r(P.candB.q); t.w(P.cand.q); t.r(P.depth.r);
                break;
            case IK_TRACE_RAYS:
                t.r(P.hit.p0); t.w(P.cand.q); t.w(P.result.rgba);
                t.add(P.blockSteps, true, true);
                if (merged) { t.r(P.candB.q); t.r(P.depth.r); }
                if (acc) t.w(P.accumulated.rgba);
                if (P.kwWeightRecalc) t.w(P.temporal.p0);
                if (ctx->launchingHeld && P.topCount > 0) t.r(P.top);
                break;
            case IK_ACCUMULATE:
                t.r(P.result.rgba); t.w(P.accumulated.rgba);
                break;
            case IK_RIS_PRIMARY:
                t.w(P.risHit); t.w(P.rayHitPoints.rgba); t.w(P.debugOverlay.rgba);
                t.add(P.blockCost, true, true);
                break;
            case IK_RIS_CANDIDATES:
                t.r(P.risHit); t.w(P.risCand);
                break;
            case IK_RIS_TRACE_RAYS:
                t.r(P.risHit); t.r(P.risCand); t.w(P.result.rgba);
                if (acc) t.w(P.accumulated.rgba);
                break;
            default: break;
        }
    }
    ctx->traceOps.push_back(t.op);
}
void traceOrderBlocks(rt_ctx ctx, hipStream_t stream, const void* counts, const void* table, const void* table2 = nullptr,
                      const void* counts2 = nullptr, const void* table3 = nullptr, const void* table4 = nullptr) {
    if (!ctx->tracing) return;
    TraceLaunch t(ctx, stream, "order_blocks");
    t.add(counts, false, true);
    if (counts2 && counts2 != counts) t.add(counts2, false, true);
    t.w(table);
    if (table2) t.w(table2);
    if (table3) t.w(table3);
    if (table4) t.w(table4);
    ctx->traceOps.push_back(t.op);
}

// Checks the launch just made.  A failure is kept in the context (the first one wins) until finish() returns it: the
// launch of a held dispatch happens inside a later call, possibly one that cannot return a status.
void checkLaunch(rt_ctx ctx, const char* kernelName) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess || ctx->stickyRc) return;
    char buf[256];
    snprintf(buf, sizeof buf, "launch of kernel '%s' failed: %s", kernelName, hipGetErrorString(e));
    ctx->stickyRc = RT_ERR_LAUNCH;
    ctx->stickyMsg = buf;
}
// Return value of an entry point that ended well: a launch failure recorded since the last report, or RT_OK.
#define LAUNCH_TRY(ctx, name)                          \
    do {                                               \
        checkLaunch(ctx, name);                        \
        if ((ctx)->stickyRc) return finish(ctx);       \
    } while (0)
int finish(rt_ctx ctx);
int finish(rt_ctx ctx) {
    if (!ctx->stickyRc) return RT_OK;
    const int rc = ctx->stickyRc;
    ctx->err = ctx->stickyMsg;
    ctx->stickyRc = 0;
    ctx->stickyMsg.clear();
    return rc;
}
void launchHeld(rt_ctx ctx);
// The dispatch stream, for everything except the merged launches themselves: what is held back goes first.
// SK: for the frame's own kernels and for calls that put nothing on the stream the next frame's primary rays could
// conflict with.  S: anything else (marks the stream dirty, see rt_ctx_s::mainDirty).
// The stream of the frame being dispatched (no held launch first).  On a stream other than stream0, work that went
// onto stream0 meanwhile (dirty) is waited for first.
hipStream_t frameStreamNow(rt_ctx ctx) {
    const int c = ctx->cur;
    if (c >= 1) {
        if (ctx->dirty[c]) {
            if (evRecord(ctx, ctx->evTail0, ctx->stream0) == hipSuccess && evWait(ctx, ctx->frameStream[c], ctx->evTail0) == hipSuccess) ctx->dirty[c] = false;
        }
        ctx->ahead[c] = true;
    }
    return ctx->frameStream[c];
}
// stream0 behind everything the other frame streams hold.
void collapse(rt_ctx ctx) {
    // (stream1: the primary rays are covered by the frame streams, which wait for them; a table refresh behind them is not)
    if (ctx->asidePending && evWait(ctx, ctx->stream0, ctx->evTableAside) == hipSuccess) ctx->asidePending = false;
    for (int s = 1; s < MAX_FRAMES; s++) {
        if (!ctx->ahead[s]) continue;
        if (evRecord(ctx, ctx->evJoin[s], ctx->frameStream[s]) == hipSuccess && evWait(ctx, ctx->stream0, ctx->evJoin[s]) == hipSuccess) ctx->ahead[s] = false;
    }
}
bool anyPending(rt_ctx ctx) {
    for (int s = 1; s < MAX_FRAMES; s++) if (ctx->ahead[s]) return true;
    return false;
}
void markDirty(rt_ctx ctx) {
    ctx->mainDirty = true;
    for (int s = 1; s < MAX_FRAMES; s++) ctx->dirty[s] = true;
}
hipStream_t SK(rt_ctx ctx) {
    if (ctx->held.kind) launchHeld(ctx);
    return frameStreamNow(ctx);
}
hipStream_t S(rt_ctx ctx) {
    if (ctx->held.kind) launchHeld(ctx);
    collapse(ctx);
    markDirty(ctx);
    // (a host operation - a copy, a conversion, a scene rebuild - may touch anything)
    if (ctx->tracing) ctx->traceOps.push_back({'H', 0, nullptr, "host", 0, {}});
    return ctx->stream0;
}
// Back to the one dispatch stream, behind everything (the start of a frame that will not run beside the one before).
void useMainStream(rt_ctx ctx) {
    if (ctx->held.kind) launchHeld(ctx);
    collapse(ctx);
    ctx->cur = 0;
}
// Before / after a launch of the frame being dispatched that touches a single-copy object all frames in flight use.
void waitShared(rt_ctx ctx, int object, hipStream_t stream) {
    for (int s = 0; s < MAX_FRAMES; s++) {
        if (s == ctx->cur || !ctx->sharedValid[object][s]) continue;
        hipError_t e = evWait(ctx, stream, ctx->evShared[object][s]);
        (void)e;
    }
}
void markShared(rt_ctx ctx, int object, hipStream_t stream) {
    if ((ctx->framesInFlight < 2 && ctx->risFramesInFlight < 2) || ctx->streamShared) { ctx->sharedValid[object][ctx->cur] = false; return; }
    ctx->sharedValid[object][ctx->cur] = evRecord(ctx, ctx->evShared[object][ctx->cur], stream) == hipSuccess;
}
// The stream of the frame being dispatched waits for all the other frame streams hold.
void joinOthers(rt_ctx ctx) {
    hipStream_t mine = nullptr;
    for (int s = 0; s < MAX_FRAMES; s++) {
        if (s == ctx->cur || !ctx->frameStream[s]) continue;
        if (s >= 1 && !ctx->ahead[s]) continue;   // (nothing there that stream0 is not already behind)
        if (!mine) mine = frameStreamNow(ctx);
        if (evRecord(ctx, ctx->evOther[s], ctx->frameStream[s]) == hipSuccess) { hipError_t e = evWait(ctx, mine, ctx->evOther[s]); (void)e; }
    }
}
// `recordedHere`: an event the caller has just recorded at this very point of the stream (every record is a packet of
// its own between two kernels: one serves both purposes).
void markAfterCandidates(rt_ctx ctx, hipStream_t stream, hipEvent_t recordedHere = nullptr) {
    if (!ctx->overlap || !ctx->behindCandidates || !ctx->evAfterCandidates) return;
    if (recordedHere) { ctx->afterCandidatesEv = recordedHere; ctx->afterCandidatesValid = true; return; }
    ctx->afterCandidatesEv = ctx->evAfterCandidates;
    ctx->afterCandidatesValid = evRecord(ctx, ctx->evAfterCandidates, stream) == hipSuccess;
}

Planes planesOf(rt_buffer b) {
    Planes p;
    char* base = (char*)b->dev;
    p.count = b->count;
    p.p0 = (float4*)base;
    p.p1 = (float4*)(base + (size_t)b->count * 16);
    p.p2 = (float4*)(base + (size_t)b->count * 32);
    p.p3 = (float*)(base + (size_t)b->count * 48);
    return p;
}

Recs recsOf(rt_buffer b) {
    Recs r;
    r.q = (float4*)b->dev;
    r.count = b->count;
    r.tag = b->tag;
    r.tagValue = (uint32_t)b->ctx->epoch;
    return r;
}
// Device bytes of a buffer in a given layout: reservoir records are padded from 52 to 64 bytes.
size_t layoutBytes(rt_buffer b, int layout) { return layout == LAYOUT_RESERVOIR_RECS ? (size_t)b->count * 64 : b->bytes; }

// The other copies of a record buffer and their tags, for the kernels that pick a pixel's newest record.
OtherCopies otherCopiesOf(rt_buffer b) {
    OtherCopies o;
    o.n = 0;
    for (int i = 0; i < MAX_FRAMES - 1; i++) { o.q[i] = nullptr; o.tag[i] = nullptr; }
    if (!b->tag) return o;
    for (int i = 0; i < b->more.n; i++) {
        if (!b->more.alt[i] || !b->more.tag[i]) continue;
        o.q[o.n] = (const float4*)b->more.alt[i];
        o.tag[o.n] = b->more.tag[i];
        o.n++;
    }
    return o;
}
void freeCopies(rt_ctx ctx, Copies& c) {
    hipError_t e = hipSuccess;
    for (int i = 0; i < MAX_FRAMES - 1; i++) {
        if (c.alt[i]) e = devFree(ctx, c.alt[i]);
        if (c.tag[i]) e = devFree(ctx, c.tag[i]);
        c.alt[i] = nullptr; c.tag[i] = nullptr;
    }
    (void)e;
    c.n = 0;
}
// Enqueues the conversion of a 52-byte-stride buffer's contents into the ABI layout at `dst`.
void launchToAos(rt_buffer b, void* dst, int64_t first = 0, int64_t n = -1) {
    if (n < 0) n = b->count - first;
    if (n <= 0) return;
    const int blocks = (int)((n + 255) / 256);
    if (b->layout == LAYOUT_RESERVOIR_RECS) hipLaunchKernelGGL(k_recs_to_aos, dim3(blocks), dim3(256), 0, S(b->ctx), recsOf(b), otherCopiesOf(b), (float*)dst, first, n);
    else hipLaunchKernelGGL(k_planes_to_aos, dim3(blocks), dim3(256), 0, S(b->ctx), planesOf(b), (float*)dst, b->layout == LAYOUT_RESERVOIR_PLANES ? 0 : 1, first, n);
}

// A record buffer goes back to a single copy.
void dropCopies(rt_buffer b) {
    hipError_t e = hipSuccess;
    if (b->tag) e = devFree(b->ctx, b->tag);
    (void)e;
    b->tag = nullptr;
    freeCopies(b->ctx, b->more);
    auto& v = b->ctx->tagged;
    v.erase(std::remove(v.begin(), v.end(), b), v.end());
}
// Completes the current copy of records [first, first + n) of a record buffer with several copies (all of them: n < 0),
// on stream0 behind everything: afterwards that copy alone is what a single buffer would hold.
int64_t tagCount(rt_buffer b) { return b->layout == LAYOUT_RESERVOIR_RECS ? b->count : b->count / b->tagGroup; }
void mergeRecords(rt_buffer b, int64_t first = 0, int64_t n = -1) {
    if (!b->tag || b->more.n == 0) return;
    if (n < 0) n = tagCount(b) - first;
    if (n <= 0) return;
    hipStream_t st = S(b->ctx);
    hipLaunchKernelGGL(k_recs_merge, dim3((unsigned)((n * b->tagGroup + 255) / 256)), dim3(256), 0, st, recsOf(b), otherCopiesOf(b), first, n, b->tagGroup);
    hipLaunchKernelGGL(k_recs_merge_tags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, recsOf(b), otherCopiesOf(b), first, n);
}
// Before the host changes, or is given the address of, a per-pixel object the frames keep two copies of: which copy
// holds a pixel's newest record follows from where the frames found hits, and that is what the host may now change.
void settleRecords(rt_ctx ctx) {
    for (rt_buffer b : ctx->tagged) mergeRecords(b);
}
// Switches a 52-byte-stride buffer between the ABI's array-of-structures layout and an internal one.
int convertLayout(rt_buffer b, int target) {
    if (b->layout == target) return RT_OK;
    rt_ctx ctx = b->ctx;
    if (ctx->noAlloc)
        return fail(ctx, RT_ERR_NOT_PREPARED, "a per-pixel buffer is not in the layout this kernel keeps it in: converting it allocates; "
                                              "dispatch the kernel once with rt_dispatch after creating or uploading the buffer");
    if (b->stride != 52) return fail(ctx, RT_ERR_INVALID_ARG, "buffer stride %d cannot hold a 52-byte per-pixel record", b->stride);
    if (b->layout != LAYOUT_AOS && target != LAYOUT_AOS) return fail(ctx, RT_ERR_INVALID_ARG, "buffer is bound as two different record types");
    if (b->zeroed && !b->tag && b->more.n == 0) {
        // a buffer nobody has written since it was created (every buffer after a scene reload): zeros are zeros in every
        // layout, so there is nothing to convert, and nothing to wait for
        const size_t need = layoutBytes(b, target);
        if (need != layoutBytes(b, b->layout)) {
            void* fresh = nullptr;
            HIP_TRY(ctx, devAlloc(ctx, &fresh, need));
            hipError_t e = hipMemsetAsync(fresh, 0, need, S(ctx));
            if (e != hipSuccess) { hipError_t e2 = devFree(ctx, fresh); (void)e2; return fail(ctx, RT_ERR_NO_DEVICE, "clearing a buffer failed: %s", hipGetErrorString(e)); }
            HIP_TRY(ctx, devFree(ctx, b->dev));
            b->dev = fresh;
        }
        b->layout = target;
        return RT_OK;
    }
    void* tmp = nullptr;
    HIP_TRY(ctx, devAlloc(ctx, &tmp, layoutBytes(b, target)));
    int blocks = (int)((b->count + 255) / 256);
    rt_buffer_s shadow = *b;
    shadow.dev = tmp;
    shadow.tag = nullptr;
    shadow.more = Copies();
    if (target == LAYOUT_AOS) {
        launchToAos(b, tmp);
    } else if (target == LAYOUT_RESERVOIR_RECS) {
        hipLaunchKernelGGL(k_aos_to_recs, dim3(blocks), dim3(256), 0, S(ctx), (const float*)b->dev, recsOf(&shadow));
    } else {
        hipLaunchKernelGGL(k_aos_to_planes, dim3(blocks), dim3(256), 0, S(ctx), (const float*)b->dev, planesOf(&shadow), target == LAYOUT_RESERVOIR_PLANES ? 0 : 1);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(S(ctx));
    if (e != hipSuccess) {
        hipError_t e2 = devFree(ctx, tmp);
        (void)e2;
        return fail(ctx, RT_ERR_NO_DEVICE, "layout conversion failed: %s", hipGetErrorString(e));
    }
    HIP_TRY(ctx, devFree(ctx, b->dev));
    b->dev = tmp;
    if (b->tag) dropCopies(b);   // the tagged copies of the records have just been merged into `tmp`: one copy again
    else if (b->more.n && layoutBytes(b, target) != layoutBytes(b, b->layout)) freeCopies(ctx, b->more);   // copies of another size are of no use
    b->layout = target;
    return RT_OK;
}

// ---- constant table helpers ----
const std::vector<uint32_t>* findConst(rt_shader sh, const char* name) {
    auto it = sh->constants.find(name);
    return it == sh->constants.end() ? nullptr : &it->second;
}
uint32_t cWord(rt_shader sh, const char* name, int i = 0) {
    const auto* v = findConst(sh, name);
    return (v && (int)v->size() > i) ? (*v)[i] : 0u;
}
float cFloat(rt_shader sh, const char* name, int i = 0) {
    uint32_t w = cWord(sh, name, i);
    float f;
    memcpy(&f, &w, 4);
    return f;
}
// float3x3 constant from a column-major 4x4 (or a 3x3 stored the same way): out is row-major 3x3.
void cMat3(rt_shader sh, const char* name, float out[9]) {
    const auto* v = findConst(sh, name);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            uint32_t w = (v && v->size() >= 16) ? (*v)[c * 4 + r] : 0u;
            memcpy(&out[r * 3 + c], &w, 4);
        }
}
rt_buffer boundBuffer(rt_shader sh, int kernel, const char* name) {
    auto it = sh->buffers[kernel].find(name);
    return it == sh->buffers[kernel].end() ? nullptr : it->second;
}
rt_texture boundTexture(rt_shader sh, int kernel, const char* name) {
    auto it = sh->textures[kernel].find(name);
    return it == sh->textures[kernel].end() ? nullptr : it->second;
}

const int MAX_TREE_DEPTH = 96;  // deepest BVH (scene levels + mesh levels) the LDS stack is sized for

// Depth (in nodes, root = 1) of the subtree at `root`; -1 on an index out of range, a cycle or excessive depth.
// Leaves are checked against `nLeafTargets` (triangles or scene objects).  `links` = (indexA, indexB) per node; `seen`
// is shared by the threads that check the meshes of one scene (a node two of them reach is a shared subtree).
// `ext`: receives the lowest and highest node visited and the end of the highest leaf range.  `seenBase`: the walk is
// expected to stay at or above this node (seen[0] belongs to it); -2 when it does not.
struct Extent { uint32_t minNode = 0xffffffffu, maxNode = 0; int64_t leafEnd = 0; };
int subtreeDepth(const std::vector<Link>& bvs, uint32_t root, int64_t nLeafTargets, std::atomic<uint8_t>* seen, std::string& why, Extent* ext = nullptr,
                 size_t seenBase = 0) {
    struct Item { uint32_t node; int depth; };
    std::vector<Item> st;
    st.push_back({root, 1});
    int maxDepth = 0;
    while (!st.empty()) {
        Item it = st.back();
        st.pop_back();
        if (it.node >= bvs.size()) { why = "BVH child index out of range"; return -1; }
        if (it.depth > MAX_TREE_DEPTH) { why = "BVH deeper than the supported maximum"; return -1; }
        if (it.node < seenBase) { why = "the walk leaves the part it was confined to"; return -2; }
        if (seen[it.node - seenBase].exchange(2, std::memory_order_relaxed) == 2) { why = "BVH node reachable twice (cycle or shared subtree)"; return -1; }
        if (ext) { if (it.node < ext->minNode) ext->minNode = it.node; if (it.node > ext->maxNode) ext->maxNode = it.node; }
        if (it.depth > maxDepth) maxDepth = it.depth;
        const Link& bv = bvs[it.node];
        if (bv.indexA >= 0) {
            int64_t start = bv.indexA, count = bv.indexB < 0 ? -(int64_t)bv.indexB : bv.indexB;
            if (start + count > nLeafTargets) { why = "BVH leaf range exceeds its component buffer"; return -1; }
            if (ext && start + count > ext->leafEnd) ext->leafEnd = start + count;
        } else {
            int64_t a = -(int64_t)bv.indexA, b = bv.indexB < 0 ? -(int64_t)bv.indexB : bv.indexB;
            st.push_back({(uint32_t)a, it.depth + 1});
            st.push_back({(uint32_t)b, it.depth + 1});
        }
    }
    return maxDepth;
}

// Workgroups for k_hash_links: every wave ends with two atomic adds to the same two words, so no more waves than the
// nodes give work to (a thread takes 16 nodes or more; 1024 workgroups at most).
dim3 hashGrid(size_t n) { const size_t g = (n + 256 * 16 - 1) / (256 * 16); return dim3((unsigned)(g < 1 ? 1 : g > 1024 ? 1024 : g)); }

// Validates the scene buffers the traversal kernels index without bounds checks and (re)builds the derived
// device arrays: wide BVH nodes, baked triangles, per-object root entries.  Cached on the buffers' versions.
// `ris` selects the RIS project's buffer set: "Tris" (76-byte triangles with inline vertices), 32-byte materials, no
// vertex buffer (BVH_RIS_DL .../ShaderHelper.cs:7-33).
int prepareScene(rt_shader sh, int kernel, KParams& P, bool ris = false) {
    rt_ctx ctx = sh->ctx;
    rt_buffer bBVs = boundBuffer(sh, kernel, "BVs"), bTris = boundBuffer(sh, kernel, ris ? "Tris" : "Triangles"),
              bVerts = ris ? nullptr : boundBuffer(sh, kernel, "Vertices"), bObjs = boundBuffer(sh, kernel, "SceneObjects"),
              bMats = boundBuffer(sh, kernel, "Materials");
    if (ris) {
        if (!bBVs || !bTris || !bObjs || !bMats)
            return fail(ctx, RT_ERR_UNBOUND, "kernel %d needs BVs, Tris, SceneObjects and Materials bound", kernel);
        if (bBVs->stride != 32 || bTris->stride != 76 || bObjs->stride != 168 || bMats->stride != 32)
            return fail(ctx, RT_ERR_INVALID_ARG, "scene buffer strides must be BVs 32, Tris 76, SceneObjects 168, Materials 32");
        bVerts = bTris;  // stands in for the absent vertex buffer in the cache key below
    } else {
        if (!bBVs || !bTris || !bVerts || !bObjs || !bMats)
            return fail(ctx, RT_ERR_UNBOUND, "kernel %d needs BVs, Triangles, Vertices, SceneObjects and Materials bound", kernel);
        if (bBVs->stride != 32 || bTris->stride != 28 || bVerts->stride != 20 || bObjs->stride != 168 || bMats->stride != 108)
            return fail(ctx, RT_ERR_INVALID_ARG, "scene buffer strides must be BVs 32, Triangles 28, Vertices 20, SceneObjects 168, Materials 108");
    }
    const uint32_t sceneStart = cWord(sh, "SceneBVHStartIndex");
    auto matches = [&](const rt_ctx_s::Prep& c) {
        return c.valid && c.ris == ris && c.vBVs == bBVs->version && c.vTris == bTris->version && c.vVerts == bVerts->version &&
               c.vObjs == bObjs->version && c.pBVs == bBVs->dev && c.pTris == bTris->dev && c.pVerts == bVerts->dev &&
               c.pObjs == bObjs->dev && c.sceneStart == sceneStart;
    };
    rt_ctx_s::Prep* hit = nullptr;
    for (auto& c : ctx->preps) if (matches(c)) hit = &c;
    const bool fresh = hit != nullptr;
    if (!fresh) {
        if (ctx->noAlloc)
            return fail(ctx, RT_ERR_NOT_PREPARED, "the traversal arrays of the bound scene buffers are not built: building them allocates; "
                                                  "dispatch kernel %d once with rt_dispatch after uploading the scene", kernel);
        // a new entry, or the least recently used one (an entry that failed validation earlier is taken first); its
        // arrays are reused when they are large enough
        for (auto& c : ctx->preps) if (!c.valid) hit = &c;
        if (!hit && ctx->preps.size() < rt_ctx_s::MAX_PREPS) { ctx->preps.emplace_back(); hit = &ctx->preps.back(); }
        if (!hit) { hit = &ctx->preps[0]; for (auto& c : ctx->preps) if (c.lastUse < hit->lastUse) hit = &c; }
    }
    rt_ctx_s::Prep& pr = *hit;
    pr.lastUse = ++ctx->prepClock;
    if (!fresh) {
        pr.valid = false;
        pr.ris = ris;
        HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
        // What the host needs of the scene: the nodes' two index words (a quarter of the node array, extracted on the
        // device) and the scene objects.  Whether the triangles' vertex indices are in range is decided on the device.
        const size_t nNodes = (size_t)bBVs->count;
        std::vector<Link> bvs;
        std::vector<rt_scene_object> objs((size_t)bObjs->count);
        // the kernels address the 64-byte node and triangle records with 32-bit byte offsets
        if (nNodes >= ((size_t)1 << 26)) return fail(ctx, RT_ERR_LIMIT, "too many BVH nodes: %zu, the limit is 2^26 - 1", nNodes);
        if ((uint64_t)bTris->count >= ((uint64_t)1 << 26)) return fail(ctx, RT_ERR_LIMIT, "too many triangles: %lld, the limit is 2^26 - 1", (long long)bTris->count);
        if (sceneStart >= nNodes) return fail(ctx, RT_ERR_BAD_SCENE, "SceneBVHStartIndex %u is outside the BVs buffer", sceneStart);
        // A reload uploads the mesh BVHs it had before and a new scene BVH above them.  If the link words of the nodes
        // below the last validated scene BVH have the digest they had then (taken on the device: no read-back), what
        // that validation found out about the meshes still holds, and only the nodes above are read back and walked.
        rt_ctx_s::Validated& known = ctx->validated[ris ? 1 : 0];
        const size_t knownPrefix = known.nPrefix;
        bool prefixKnown = false;
        struct Scratch { rt_ctx ctx; void* p = nullptr; ~Scratch() { if (p) { hipError_t e = devFree(ctx, p); (void)e; } } } scratch{ctx};
        HIP_TRY(ctx, devAlloc(ctx, &scratch.p, nNodes * sizeof(Link) + 32));   // links, one flag word, two digest words
        uint32_t* flagDev = (uint32_t*)((char*)scratch.p + nNodes * sizeof(Link));
        unsigned long long* digestDev = (unsigned long long*)((char*)flagDev + 8);
        {
            uint32_t flag = 0;
            unsigned long long digest[2] = {0, 0};
            const bool tryPrefix = ctx->reuseValidation && known.valid && knownPrefix > 0 && knownPrefix <= nNodes && known.links.size() >= knownPrefix;
            hipError_t e = hipMemsetAsync(flagDev, 0, 32, S(ctx));
            hipLaunchKernelGGL(k_extract_links, dim3((unsigned)((nNodes + 255) / 256)), dim3(256), 0, S(ctx), (const rt_bv*)bBVs->dev, bBVs->count, (int2*)scratch.p);
            if (!ris)
                hipLaunchKernelGGL(k_check_tri_indices, dim3((unsigned)((bTris->count + 255) / 256)), dim3(256), 0, S(ctx),
                                   (const rt_triangle*)bTris->dev, bTris->count, bVerts->count, flagDev);
            if (tryPrefix) hipLaunchKernelGGL(k_hash_links, hashGrid(knownPrefix), dim3(256), 0, S(ctx), (const int2*)scratch.p, (int64_t)knownPrefix, digestDev);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess && tryPrefix) e = hipMemcpyAsync(digest, digestDev, 16, hipMemcpyDeviceToHost, S(ctx));
            if (e == hipSuccess) e = hipMemcpyAsync(&flag, flagDev, 4, hipMemcpyDeviceToHost, S(ctx));
            if (e == hipSuccess) e = hipMemcpyAsync(objs.data(), bObjs->dev, objs.size() * sizeof(rt_scene_object), hipMemcpyDeviceToHost, S(ctx));
            if (e == hipSuccess) e = hipStreamSynchronize(S(ctx));
            prefixKnown = tryPrefix && e == hipSuccess && digest[0] == known.digest[0] && digest[1] == known.digest[1];
            size_t first = 0;
            if (prefixKnown) { bvs = std::move(known.links); first = knownPrefix; }   // the host's copy of the lower part is still right
            known.valid = false;   // (until this preparation has succeeded)
            known.links.clear();
            if (bvs.capacity() < nNodes) bvs.reserve(nNodes + nNodes / 8 + 1024);   // (room for the scene BVHs later reloads append)
            bvs.resize(nNodes);
            if (e == hipSuccess && nNodes > first) {
                e = hipMemcpyAsync((void*)(bvs.data() + first), (const char*)scratch.p + first * sizeof(Link), (nNodes - first) * sizeof(Link), hipMemcpyDeviceToHost, S(ctx));
                if (e == hipSuccess) e = hipStreamSynchronize(S(ctx));
            }
            if (e != hipSuccess) return fail(ctx, RT_ERR_NO_DEVICE, "reading the scene back for validation failed: %s", hipGetErrorString(e));
            if (flag) return fail(ctx, RT_ERR_BAD_SCENE, "a triangle's vertex index is outside the Vertices buffer");
        }
        std::vector<int> meshRoots;
        {
            std::map<int, int> seenRoots;
            for (const auto& o : objs) {
                if (o.bvStartIndex < 0) return fail(ctx, RT_ERR_BAD_SCENE, "scene object with negative bvStartIndex");
                if (seenRoots.emplace(o.bvStartIndex, 0).second) meshRoots.push_back(o.bvStartIndex);
            }
        }
        std::string why;
        int sceneDepth = -1, maxMeshDepth = 0;
        bool reused = false;
        if (prefixKnown) {
            // the scene BVH alone, which must lie above the known part; every mesh an object names must be one of those
            // validated together last time, lie inside the known part, and find its triangles in the buffer bound now
            std::unique_ptr<std::atomic<uint8_t>[]> seenAbove(new std::atomic<uint8_t>[nNodes - knownPrefix + 1]());
            const int d = subtreeDepth(bvs, sceneStart, bObjs->count, seenAbove.get(), why, nullptr, knownPrefix);
            if (d == -1) return fail(ctx, RT_ERR_BAD_SCENE, "scene BVH (SceneBVHStartIndex %u): %s", sceneStart, why.c_str());
            bool ok = d >= 0;
            for (size_t k = 0; ok && k < meshRoots.size(); k++) {
                const auto it = known.meshes.find((uint32_t)meshRoots[k]);
                ok = it != known.meshes.end() && (size_t)meshRoots[k] < knownPrefix && it->second.maxNode < knownPrefix && it->second.leafEnd <= bTris->count;
                if (ok && it->second.depth > maxMeshDepth) maxMeshDepth = it->second.depth;
            }
            if (ok) { reused = true; sceneDepth = d; }
            else maxMeshDepth = 0;
        }
        if (!reused) {
            std::unique_ptr<std::atomic<uint8_t>[]> seen(new std::atomic<uint8_t>[nNodes]());
            Extent sceneExt;
            sceneDepth = subtreeDepth(bvs, sceneStart, bObjs->count, seen.get(), why, &sceneExt);
            if (sceneDepth < 0) return fail(ctx, RT_ERR_BAD_SCENE, "scene BVH (SceneBVHStartIndex %u): %s", sceneStart, why.c_str());
            // the mesh BVHs, each checked once, on as many threads as the machine offers (at most 16): a 10 M-triangle
            // scene of distinct meshes has 19 M nodes to walk
            std::vector<int> meshDepths(meshRoots.size(), 0);
            std::vector<std::string> meshWhy(meshRoots.size());
            std::vector<Extent> meshExt(meshRoots.size());
            {
                unsigned nThreads = std::thread::hardware_concurrency();
                nThreads = nThreads < 1 ? 1 : nThreads > 16 ? 16 : nThreads;
                if (nThreads > meshRoots.size()) nThreads = (unsigned)meshRoots.size();
                std::atomic<size_t> next(0);
                auto work = [&] {
                    for (size_t k; (k = next.fetch_add(1)) < meshRoots.size();)
                        meshDepths[k] = subtreeDepth(bvs, (uint32_t)meshRoots[k], bTris->count, seen.get(), meshWhy[k], &meshExt[k]);
                };
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < nThreads; t++) pool.emplace_back(work);
                work();
                for (auto& t : pool) t.join();
            }
            for (size_t k = 0; k < meshRoots.size(); k++) {
                if (meshDepths[k] < 0) return fail(ctx, RT_ERR_BAD_SCENE, "mesh BVH at %d: %s", meshRoots[k], meshWhy[k].c_str());
                if (meshDepths[k] > maxMeshDepth) maxMeshDepth = meshDepths[k];
            }
            // what the next preparation may rely on: the meshes that lie wholly below the scene BVH, and the digest of that part
            known.meshes.clear();
            known.nPrefix = sceneExt.minNode <= nNodes ? sceneExt.minNode : 0;
            for (size_t k = 0; k < meshRoots.size(); k++)
                if (meshExt[k].maxNode < known.nPrefix) known.meshes[(uint32_t)meshRoots[k]] = {meshDepths[k], meshExt[k].maxNode, meshExt[k].leafEnd};
            known.digest[0] = known.digest[1] = 0;
            if (known.nPrefix > 0 && !known.meshes.empty()) {
                hipError_t e = hipMemsetAsync(digestDev, 0, 16, S(ctx));
                hipLaunchKernelGGL(k_hash_links, hashGrid(known.nPrefix), dim3(256), 0, S(ctx), (const int2*)scratch.p, (int64_t)known.nPrefix, digestDev);
                if (e == hipSuccess) e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(known.digest, digestDev, 16, hipMemcpyDeviceToHost, S(ctx));
                if (e == hipSuccess) e = hipStreamSynchronize(S(ctx));
                if (e != hipSuccess) known.nPrefix = 0;
            } else known.nPrefix = 0;
        }
        if (ctx->tracing) {   // (for tests: which way this preparation went)
            TraceLaunch t(ctx, ctx->stream0, reused ? "scene_check_reused" : "scene_check_full");
            ctx->traceOps.push_back(t.op);
        }
        pr.stackDepth = sceneDepth + maxMeshDepth;   // entries stacked on the deepest path (one per internal level) + 2
        // the per-lane stack lives in LDS: 4 waves x 64 lanes x 4 bytes per entry, within the 64 KiB a workgroup may ask for
        if ((size_t)WAVES_PER_BLOCK * pr.stackDepth * 64 * 4 > 65536)
            return fail(ctx, RT_ERR_LIMIT, "BVH too deep: scene BVH of %d levels + mesh BVH of %d levels need a traversal stack of %d entries, the limit is %d",
                        sceneDepth, maxMeshDepth, pr.stackDepth, 65536 / (WAVES_PER_BLOCK * 64 * 4));

        auto ensure = [&](void** p, size_t* have, size_t need) -> int {
            if (*have >= need && *p) return RT_OK;
            if (*p) { hipError_t e = devFree(ctx, *p); (void)e; *p = nullptr; *have = 0; }
            HIP_TRY(ctx, devAlloc(ctx, p, need));
            *have = need;
            return RT_OK;
        };
        int rc;
        if ((rc = ensure((void**)&pr.wide, &pr.wideBytes, (size_t)bBVs->count * 64))) return rc;
        if ((rc = ensure((void**)&pr.btri, &pr.btriBytes, (size_t)bTris->count * 64))) return rc;
        if ((rc = ensure((void**)&pr.objRoot, &pr.objRootBytes, (size_t)bObjs->count * 4))) return rc;
        hipLaunchKernelGGL(k_build_wide, dim3((unsigned)((bBVs->count + 255) / 256)), dim3(256), 0, S(ctx),
                           (const rt_bv*)bBVs->dev, bBVs->count, pr.wide);
        if (ris)
            hipLaunchKernelGGL(k_ris_bake_tris, dim3((unsigned)((bTris->count + 255) / 256)), dim3(256), 0, S(ctx),
                               (const rt_ris_tri*)bTris->dev, bTris->count, pr.btri);
        else
            hipLaunchKernelGGL(k_bake_tris, dim3((unsigned)((bTris->count + 255) / 256)), dim3(256), 0, S(ctx),
                               (const rt_triangle*)bTris->dev, bTris->count, (const rt_vertex*)bVerts->dev, bVerts->count, pr.btri);
        hipLaunchKernelGGL(k_obj_root_entries, dim3((unsigned)((bObjs->count + 255) / 256)), dim3(256), 0, S(ctx),
                           (const rt_bv*)bBVs->dev, bBVs->count, (const rt_scene_object*)bObjs->dev, bObjs->count, pr.objRoot);
        LAUNCH_TRY(ctx, "build_wide / bake_tris / obj_root_entries");
        // stack entry of the scene root, same rule as the device-side entryOf()
        {
            const Link& r = bvs[sceneStart];
            if (r.indexA >= 0) {
                if (r.indexB >= 0 && r.indexB <= ENTRY_INLINE_MAX_COUNT && r.indexA <= ENTRY_INLINE_MAX_START)
                    pr.sceneRootEntry = ENTRY_LEAF | ((uint32_t)r.indexB << 26) | (uint32_t)r.indexA;
                else pr.sceneRootEntry = ENTRY_LEAF_INDIRECT | sceneStart;
            } else pr.sceneRootEntry = sceneStart;
        }
        // The table of top nodes, when asked for: breadth first from the scene root, through the scene-level leaves into
        // the objects' mesh BVHs (each mesh once), internal nodes only, until the table is full.
        pr.topCount = 0;
        pr.sceneRootEntryTop = pr.sceneRootEntry;
        if (ctx->topNodes > 0 && !ris) {
            TopPlan plan;
            memset(&plan, 0, sizeof plan);
            const int want = ctx->topNodes > TOP_MAX ? TOP_MAX : ctx->topNodes;
            struct Item { uint32_t node; bool sceneLevel; int parentSlot; int side; };
            std::vector<Item> queue;
            std::map<uint32_t, int> slotOfRoot;   // mesh roots already queued
            auto internal = [&](uint32_t n) { return n < nNodes && bvs[n].indexA < 0; };
            if (internal(sceneStart)) queue.push_back({sceneStart, true, -1, 0});
            else if (bvs[sceneStart].indexA >= 0) {   // the scene root is a leaf: straight into its objects
                const int64_t start = bvs[sceneStart].indexA, count = bvs[sceneStart].indexB < 0 ? -(int64_t)bvs[sceneStart].indexB : bvs[sceneStart].indexB;
                for (int64_t k = start; k < start + count && k < (int64_t)objs.size(); k++) {
                    const uint32_t r = (uint32_t)objs[(size_t)k].bvStartIndex;
                    if (internal(r) && slotOfRoot.emplace(r, 0).second) queue.push_back({r, false, -1, 0});
                }
            }
            for (size_t head = 0; head < queue.size() && plan.n < want; head++) {
                const Item it = queue[head];
                const int slot = plan.n++;
                plan.node[slot] = it.node;
                plan.slotA[slot] = plan.slotB[slot] = -1;
                if (it.parentSlot >= 0) (it.side == 0 ? plan.slotA : plan.slotB)[it.parentSlot] = (int16_t)slot;
                const Link& bv = bvs[it.node];
                const uint32_t child[2] = {(uint32_t)(-(int64_t)bv.indexA), (uint32_t)(bv.indexB < 0 ? -(int64_t)bv.indexB : (int64_t)bv.indexB)};
                for (int side = 0; side < 2; side++) {
                    const uint32_t c = child[side];
                    if (c >= nNodes) continue;
                    if (bvs[c].indexA < 0) queue.push_back({c, it.sceneLevel, slot, side});
                    else if (it.sceneLevel) {   // a scene-level leaf: the mesh roots of its objects join the queue (no parent record names them)
                        const int64_t start = bvs[c].indexA, count = bvs[c].indexB < 0 ? -(int64_t)bvs[c].indexB : bvs[c].indexB;
                        for (int64_t k = start; k < start + count && k < (int64_t)objs.size(); k++) {
                            const uint32_t r = (uint32_t)objs[(size_t)k].bvStartIndex;
                            if (internal(r) && slotOfRoot.emplace(r, 0).second) queue.push_back({r, false, -1, 0});
                        }
                    }
                }
            }
            // (children queued after the table filled up have no slot: their parents keep the BV index, slotA/B stay -1;
            // a child's slot was set when the child itself was taken off the queue)
            if (plan.n > 0) {
                if ((rc = ensure((void**)&pr.top, &pr.topBytes, (size_t)TOP_MAX * 64))) return rc;
                if ((rc = ensure((void**)&pr.objRootTop, &pr.objRootTopBytes, (size_t)bObjs->count * 4))) return rc;
                hipLaunchKernelGGL(k_build_top, dim3(1), dim3(TOP_MAX), 0, S(ctx), (const float4*)pr.wide, plan, pr.top);
                hipLaunchKernelGGL(k_obj_root_entries_top, dim3((unsigned)((bObjs->count + 255) / 256)), dim3(256), 0, S(ctx),
                                   (const uint32_t*)pr.objRoot, (const rt_scene_object*)bObjs->dev, bObjs->count, plan, pr.objRootTop);
                LAUNCH_TRY(ctx, "build_top / obj_root_entries_top");
                pr.topCount = plan.n;
                if (plan.node[0] == sceneStart && internal(sceneStart)) pr.sceneRootEntryTop = ENTRY_TOP | 0u;
            }
        }
        pr.vBVs = bBVs->version; pr.vTris = bTris->version; pr.vVerts = bVerts->version; pr.vObjs = bObjs->version;
        pr.pBVs = bBVs->dev; pr.pTris = bTris->dev; pr.pVerts = bVerts->dev; pr.pObjs = bObjs->dev;
        pr.sceneStart = sceneStart;
        pr.valid = true;
        if (known.nPrefix > 0) { known.links = std::move(bvs); known.valid = true; }
    }
    P.wide = pr.wide; P.btri = pr.btri; P.objRootEntry = pr.objRoot;
    P.BVs = (const rt_bv*)bBVs->dev; P.nBVs = bBVs->count;
    P.objs = (const rt_scene_object*)bObjs->dev; P.nObjs = bObjs->count;
    P.nTris = bTris->count; P.nMats = bMats->count;
    if (ris) {
        P.risTris = (const rt_ris_tri*)bTris->dev;
        P.risMats = (const rt_ris_material*)bMats->dev;
    } else {
        P.tris = (const rt_triangle*)bTris->dev;
        P.verts = (const rt_vertex*)bVerts->dev; P.nVerts = bVerts->count;
        P.mats = (const rt_material*)bMats->dev;
    }
    P.sceneRootEntry = pr.sceneRootEntry;
    P.stackDepth = pr.stackDepth;
    P.top = pr.top; P.topCount = pr.topCount; P.objRootEntryTop = pr.objRootTop; P.sceneRootEntryTop = pr.sceneRootEntryTop;
    return RT_OK;
}

int needPlanes(rt_shader sh, int kernel, const char* name, int layout, Planes& out) {
    rt_buffer b = boundBuffer(sh, kernel, name);
    if (!b) return fail(sh->ctx, RT_ERR_UNBOUND, "kernel %d: buffer '%s' is not bound", kernel, name);
    int rc = convertLayout(b, layout);
    if (rc) return rc;
    b->zeroed = false;   // (the kernel may write it)
    out = planesOf(b);
    return RT_OK;
}
// `duringFrame`: the kernel runs after the frame's InitialTrace; a buffer that dispatch did not give a copy of its own for
// this frame may still be in use by the other frame in flight.
int needRecs(rt_shader sh, int kernel, const char* name, Recs& out, bool duringFrame = true) {
    rt_buffer b = boundBuffer(sh, kernel, name);
    if (!b) return fail(sh->ctx, RT_ERR_UNBOUND, "kernel %d: buffer '%s' is not bound", kernel, name);
    int rc = convertLayout(b, LAYOUT_RESERVOIR_RECS);
    if (rc) return rc;
    b->zeroed = false;
    if (duringFrame && b->epoch != sh->ctx->epoch) joinOthers(sh->ctx);
    out = recsOf(b);
    return RT_OK;
}
int needTex(rt_shader sh, int kernel, const char* name, bool r32, int resX, int resY, Tex& out) {
    rt_texture t = boundTexture(sh, kernel, name);
    if (!t) return fail(sh->ctx, RT_ERR_UNBOUND, "kernel %d: texture '%s' is not bound", kernel, name);
    if ((t->format == RT_FORMAT_R32F) != r32) return fail(sh->ctx, RT_ERR_INVALID_ARG, "texture '%s' has the wrong format", name);
    if (t->w != resX || t->h != resY) return fail(sh->ctx, RT_ERR_INVALID_ARG, "texture '%s' is %dx%d but Resolution is %dx%d", name, t->w, t->h, resX, resY);
    out.w = t->w; out.h = t->h;
    out.rgba = r32 ? nullptr : (float4*)t->dev;
    out.r = r32 ? (float*)t->dev : nullptr;
    out.unorm8 = t->format == RT_FORMAT_RGBA8_UNORM;
    return RT_OK;
}
void optionalTex(rt_shader sh, int kernel, const char* name, Tex& out) {
    rt_texture t = boundTexture(sh, kernel, name);
    out.rgba = nullptr; out.r = nullptr; out.w = out.h = 0; out.unorm8 = 0;
    if (t && t->format != RT_FORMAT_R32F) { out.rgba = (float4*)t->dev; out.w = t->w; out.h = t->h; }
}

void fillConstants(rt_shader sh, KParams& P) {
    P.MaxBVHDepth = cWord(sh, "MaxBVHDepth");
    P.ScatterProbability = cFloat(sh, "ScatterProbability");
    P.MaxBounceCount = cWord(sh, "MaxBounceCount");
    P.ResX = (int32_t)cWord(sh, "Resolution", 0);
    P.ResY = (int32_t)cWord(sh, "Resolution", 1);
    P.ViewSpaceDimsX = cFloat(sh, "ViewSpaceDims", 0);
    P.ViewSpaceDimsY = cFloat(sh, "ViewSpaceDims", 1);
    P.DefocusStrength = cFloat(sh, "DefocusStrength");
    P.FocalPlaneFactor = cFloat(sh, "FocalPlaneFactor");
    for (int i = 0; i < 3; i++) { P.camPos[i] = cFloat(sh, "WorldSpaceCameraPos", i); P.lastCamPos[i] = cFloat(sh, "LastWorldSpaceCameraPos", i); }
    cMat3(sh, "CameraTransform", P.camT);
    cMat3(sh, "LastCameraTransformInverse", P.lastCamTInv);
    P.FrameRand = cWord(sh, "FrameRand");
    P.FrameCount = cWord(sh, "FrameCount");
    P.DebugMaxX = (int32_t)cWord(sh, "DebugDataMaxValues", 0);
    P.DebugMaxY = (int32_t)cWord(sh, "DebugDataMaxValues", 1);
    P.SceneObjectReservoirTestsNum = cWord(sh, "SceneObjectReservoirTestsNum");
    P.TriReservoirTestsNum = cWord(sh, "TriReservoirTestsNum");
    P.ReuseBufferCycle = (int32_t)cWord(sh, "ReuseBufferCycle");
    P.MaxOffset = (int32_t)cWord(sh, "MaxOffset");
    P.TemporalCandidatesNum = cWord(sh, "TemporalCandidatesNum");
    P.SpatialHitPointDiffThreshold = cFloat(sh, "SpatialHitPointDiffThreshold");
    P.SpatialNormalsAngleThreshold = cFloat(sh, "SpatialNormalsAngleThreshold");
    P.SpatialBRDFThreshold = cFloat(sh, "SpatialBRDFThreshold");
    P.TemporalReuseWeight = cFloat(sh, "TemporalReuseWeight");
    P.PixelMovementThreshold = cFloat(sh, "PixelMovementThreshold");
    P.TemporalPrecisionThreshold = cFloat(sh, "TemporalPrecisionThreshold");
    P.EmittingObjectsNum = cWord(sh, "EmittingObjectsNum");
    P.atlasDimsX = cFloat(sh, "TextureAtlasDims", 0);
    P.atlasDimsY = cFloat(sh, "TextureAtlasDims", 1);
    P.envDimsX = cFloat(sh, "EnvironmentMapTexDims", 0);
    P.envDimsY = cFloat(sh, "EnvironmentMapTexDims", 1);
    P.kwVisReuse = sh->kwVisReuse; P.kwWeightRecalc = sh->kwWeightRecalc; P.kwBrdf = sh->kwBrdf;
    P.ppFrameCount = (int32_t)cWord(sh, "FrameCount");
}

// Order in which workgroups render the bx x by grid of 16x16-pixel blocks.  Workgroup i of a launch runs on XCD
// i mod 8 and workgroups start in index order, so the table is built as 8 per-XCD lists interleaved: each XCD
// walks compact regions of the image (its L2 keeps seeing the same part of the BVH), regions are dealt to the
// XCDs cyclically (cost varies a lot over the image, so contiguous eighths would leave most XCDs idle), and at any
// moment the 8 XCDs work on neighbouring regions of similar cost.
// mode 0: plain row-major order; 1: one contiguous eighth of the image per XCD; 2 (default): cyclic regions.
std::vector<uint32_t> buildBlockMap(unsigned bx, unsigned by, int mode) {
    std::vector<uint32_t> rowMajor;
    for (unsigned y = 0; y < by; y++) for (unsigned x = 0; x < bx; x++) rowMajor.push_back(y << 16 | x);
    if (mode == 0) return rowMajor;
    std::vector<std::vector<uint32_t>> lists(8);
    if (mode == 1) {
        size_t n = rowMajor.size(), q = n / 8, r = n % 8, pos = 0;
        for (unsigned x = 0; x < 8; x++) {
            size_t len = q + (x < r ? 1 : 0);
            lists[x].assign(rowMajor.begin() + pos, rowMajor.begin() + pos + len);
            pos += len;
        }
    } else {
        const unsigned RW = RT_REGION, RH = RT_REGION;  // region = 4x4 blocks = 64x64 pixels
        unsigned rx = (bx + RW - 1) / RW, ry = (by + RH - 1) / RH, region = 0;
        for (unsigned j = 0; j < ry; j++)
            for (unsigned i = 0; i < rx; i++, region++) {
                // shift the cyclic assignment by row so vertically adjacent regions go to different XCDs too
                std::vector<uint32_t>& l = lists[(region + 3 * j) % 8];
                for (unsigned y = j * RH; y < (j + 1) * RH && y < by; y++)
                    for (unsigned x = i * RW; x < (i + 1) * RW && x < bx; x++) l.push_back(y << 16 | x);
            }
    }
    size_t longest = 0;
    for (auto& l : lists) longest = l.size() > longest ? l.size() : longest;
    std::vector<uint32_t> out;
    out.reserve(longest * 8);
    for (size_t k = 0; k < longest; k++)
        for (unsigned x = 0; x < 8; x++) out.push_back(k < lists[x].size() ? lists[x][k] : 0xffffffffu);
    while (out.size() >= 8) {  // drop trailing rounds that are all padding
        bool allPad = true;
        for (size_t i = out.size() - 8; i < out.size(); i++) allPad = allPad && out[i] == 0xffffffffu;
        if (!allPad) break;
        out.resize(out.size() - 8);
    }
    return out;
}

// Launch geometry of the per-pixel kernels: 256-thread workgroups, each a 2x2 group of 8x8 tiles, covering the
// context's row band (whole image by default).  The caller's group counts only have to cover the image.
int pixelGrid(rt_ctx ctx, KParams& P, int gx, int gy, unsigned* blocks, rt_ctx_s::BlockMap** used = nullptr) {
    if (P.ResX <= 0 || P.ResY <= 0) return fail(ctx, RT_ERR_UNBOUND, "constant 'Resolution' is not set");
    if ((int64_t)gx * 8 < P.ResX || (int64_t)gy * 8 < P.ResY)
        return fail(ctx, RT_ERR_INVALID_ARG, "dispatch of %dx%d groups does not cover Resolution %dx%d", gx, gy, P.ResX, P.ResY);
    P.row0 = 0; P.row1 = P.ResY;
    if (ctx->row1 > 0) { P.row0 = ctx->row0 < 0 ? 0 : ctx->row0; P.row1 = ctx->row1 < P.ResY ? ctx->row1 : P.ResY; }
    if (P.row1 <= P.row0) { *blocks = 0; return RT_OK; }
    int bandY0 = P.row0 & ~7;
    unsigned bx = (unsigned)(P.ResX + 15) / 16, by = (unsigned)(P.row1 - bandY0 + 15) / 16;
    if (bx > 0xffffu || by > 0xffffu) return fail(ctx, RT_ERR_LIMIT, "image too large for the block table");
    static const int mode = [] { const char* e = getenv("RT_BLOCK_MAP"); return e ? atoi(e) : 2; }();
    static const bool reorder = [] { const char* e = getenv("RT_BLOCK_ORDER"); return e ? atoi(e) != 0 : true; }();
    P.blocksX = (int)bx;
    P.blockCost = nullptr;
    if (ctx->noAlloc) {
        // The table is built in the file-scope scratch, on the stream, for this launch alone: hit counts cleared, then
        // k_order_blocks on all-zero counts deals the regions to the XCDs in some order (any order renders the same image).
        const unsigned regions = ((bx + RT_REGION - 1) / RT_REGION) * ((by + RT_REGION - 1) / RT_REGION);
        if (regions > NOALLOC_MAX_REGIONS)
            return fail(ctx, RT_ERR_LIMIT, "rt_dispatch_no_alloc: %u x %u pixels are %u regions of 64x64 pixels, the scratch table holds %u",
                        bx * 16, by * 16, regions, NOALLOC_MAX_REGIONS);
        rt_ctx_s::BlockMap& m = ctx->noAllocMap;
        m.bx = (int)bx; m.by = (int)by;
        m.dev = m.dyn = ctx->scratch;
        m.cost = ctx->scratch + NOALLOC_TABLE_WORDS;
        m.n = m.nDyn = (regions + 7) / 8 * (8 * RT_REGION * RT_REGION);
        m.dynValid = false; m.age = 0;
        HIP_TRY(ctx, hipMemsetAsync(m.cost, 0, (size_t)bx * by * 4 * sizeof(uint32_t), S(ctx)));
        m.steps = nullptr; m.stepsValid = false; m.rankedBySteps = false; m.dynHits = nullptr; m.pendingDyn = false; m.pendingMeasured = false;
        m.rs = m.rsHits = m.rsRays = RT_REGION; m.nDynHits = 0; m.dynRays = m.dyn; m.nDynRays = m.nDyn;   // one table for every kernel
        m.dynSpare = m.dynRaysSpare = nullptr; m.swapChange = 0; m.pendingSwap = false;
        OrderTables one = {};
        one.t[0].dyn = m.dyn; one.t[0].nDyn = m.nDyn; one.t[0].rs = RT_REGION; one.t[0].costW = m.cost; one.t[0].byMax = 0;
        hipLaunchKernelGGL(k_order_blocks, dim3(1), dim3(ORDER_THREADS), orderLdsBytes(regions), S(ctx), m.bx, m.by, one, regions);
        LAUNCH_TRY(ctx, "order_blocks");
        P.blockMap = m.dev;
        *blocks = m.n;
        if (used) *used = &m;
        return RT_OK;
    }
    auto use = [&](rt_ctx_s::BlockMap& m) {
        P.blockMap = m.dynValid ? m.dyn : m.dev;
        *blocks = m.dynValid ? m.nDyn : m.n;
        if (used) *used = &m;
        return RT_OK;
    };
    for (auto& m : ctx->blockMaps)
        if (m.bx == (int)bx && m.by == (int)by) return use(m);
    if (ctx->blockMaps.size() >= 16) {  // bounded: forget the oldest table once nothing can still be reading it
        HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
        rt_ctx_s::BlockMap& old = ctx->blockMaps.front();
        hipError_t e = hipFree(old.dev);
        if (old.dyn) e = hipFree(old.dyn);
        if (old.cost) e = hipFree(old.cost);
        if (old.steps) e = hipFree(old.steps);
        if (old.dynHits) e = hipFree(old.dynHits);
        if (old.dynRays && old.dynRays != old.dyn) e = hipFree(old.dynRays);
        if (old.dynRaysSpare && old.dynRaysSpare != old.dynSpare) e = hipFree(old.dynRaysSpare);
        if (old.dynSpare) e = hipFree(old.dynSpare);
        (void)e;
        ctx->blockMaps.erase(ctx->blockMaps.begin());
    }
    std::vector<uint32_t> map = buildBlockMap(bx, by, mode);
    rt_ctx_s::BlockMap m = {(int)bx, (int)by, nullptr, (unsigned)map.size(), nullptr, nullptr, 0, false, 0, nullptr, false, false, nullptr, false, false};
    HIP_TRY(ctx, hipMalloc((void**)&m.dev, map.size() * 4));
    HIP_TRY(ctx, hipMemcpy(m.dev, map.data(), map.size() * 4, hipMemcpyHostToDevice));
    // a table's region size: the one asked for, doubled until the ordering kernel can hold the regions (0 = it cannot)
    auto regionsAt = [&](unsigned rs) { return ((bx + rs - 1) / rs) * ((by + rs - 1) / rs); };
    auto fit = [&](int rs) { while (rs <= 8 && regionsAt((unsigned)rs) > (unsigned)ORDER_MAX_REGIONS) rs *= 2; return rs <= 8 ? rs : 0; };
    auto entries = [&](int rs) { return (regionsAt((unsigned)rs) + 7) / 8 * (unsigned)(8 * rs * rs); };
    m.rs = fit(ctx->regionBlocks[0]); m.rsHits = fit(ctx->regionBlocks[1]); m.rsRays = fit(ctx->regionBlocks[2]);
    m.nDynHits = 0; m.dynRays = nullptr; m.nDynRays = 0; m.dynSpare = m.dynRaysSpare = nullptr; m.swapChange = 0; m.pendingSwap = false;
    if (reorder && mode == 2 && m.rs && m.rsHits && m.rsRays) {
        m.nDyn = entries(m.rs);
        HIP_TRY(ctx, hipMalloc((void**)&m.dyn, (size_t)m.nDyn * 4));
        HIP_TRY(ctx, hipMalloc((void**)&m.cost, (size_t)bx * by * 4 * 4));
        HIP_TRY(ctx, hipMemset(m.cost, 0, (size_t)bx * by * 4 * 4));
        m.nDynHits = entries(m.rsHits);
        HIP_TRY(ctx, hipMalloc((void**)&m.dynHits, (size_t)m.nDynHits * 4));
        HIP_TRY(ctx, hipMalloc((void**)&m.dynSpare, (size_t)m.nDyn * 4));
        if (m.rsRays == m.rs) { m.dynRays = m.dyn; m.nDynRays = m.nDyn; m.dynRaysSpare = m.dynSpare; }
        else {
            m.nDynRays = entries(m.rsRays);
            HIP_TRY(ctx, hipMalloc((void**)&m.dynRays, (size_t)m.nDynRays * 4));
            HIP_TRY(ctx, hipMalloc((void**)&m.dynRaysSpare, (size_t)m.nDynRays * 4));
        }
        HIP_TRY(ctx, hipMalloc((void**)&m.steps, (size_t)bx * by * 4 * 4));
        HIP_TRY(ctx, hipMemset(m.steps, 0, (size_t)bx * by * 4 * 4));
    }
    ctx->blockMaps.push_back(m);
    return use(ctx->blockMaps.back());
}
void freeBlockMaps(rt_ctx ctx) {
    hipError_t e = hipSuccess;
    for (auto& m : ctx->blockMaps) {
        e = hipFree(m.dev);
        if (m.dyn) e = hipFree(m.dyn);
        if (m.cost) e = hipFree(m.cost);
        if (m.steps) e = hipFree(m.steps);
        if (m.dynHits) e = hipFree(m.dynHits);
        if (m.dynRays && m.dynRays != m.dyn) e = hipFree(m.dynRays);
        if (m.dynRaysSpare && m.dynRaysSpare != m.dynSpare) e = hipFree(m.dynRaysSpare);
        if (m.dynSpare) e = hipFree(m.dynSpare);
    }
    (void)e;
    ctx->blockMaps.clear();
}
void swapTables(rt_ctx_s::BlockMap* m) {
    const bool oneTable = m->dynRays == m->dyn;
    std::swap(m->dyn, m->dynSpare);
    if (oneTable) { m->dynRays = m->dyn; m->dynRaysSpare = m->dynSpare; }
    else std::swap(m->dynRays, m->dynRaysSpare);
}
// The tables a refresh built aside (refreshTables) come into use: from here on dispatches walk them, and every frame
// stream first waits for the launch that wrote them (long done: it ran a frame ago).  `done`: that has been waited for.
void applyTableSwap(rt_ctx ctx, rt_ctx_s::BlockMap* m, bool done = false) {
    if (!m || !m->pendingSwap) { if (m && done) { swapTables(m); m->swapChange = ctx->changes; } return; }
    m->pendingSwap = false;
    swapTables(m);
    m->swapChange = ctx->changes;
    if (done) return;
    for (int k = 0; k < MAX_FRAMES; k++) {
        if (!ctx->frameStream[k]) continue;
        hipError_t e = evWait(ctx, ctx->frameStream[k], ctx->evTableAside);
        (void)e;
    }
    ctx->asidePending = false;   // (stream0 is among them)
}
unsigned regionsOf(const rt_ctx_s::BlockMap* m, int rs) {
    return (unsigned)((m->bx + rs - 1) / rs) * (unsigned)((m->by + rs - 1) / rs);
}
// The kernels that cast secondary rays walk a table of their own (smaller regions: see rt_ctx_s::regionBlocks).  Every
// table of a grid lists the same blocks, so dispatches merged into one launch may have asked for different ones.
void useRaysTable(const rt_ctx_s::BlockMap* m, KParams& P, unsigned* blocks) {
    if (m && m->dynValid && m->dynRays) { P.blockMap = m->dynRays; *blocks = m->nDynRays; }
}
// Ranks the regions anew and rebuilds the grid's heaviest-first tables (used by every later launch on the same grid).
// Called from the InitialTrace / GenerateCandidates dispatch once the primary rays are on their stream and the frame's
// stream has been told to wait for them.  The ranking moves slowly from frame to frame: it is refreshed on the first
// frame, on the first frame after TraceRays has measured its steps, and then every `tableRefreshFrames`-th dispatch (a
// stale order is still a complete one; it only costs a little balance after a camera cut).
//
// Beside the frames in flight (`beside`: the primary rays run on stream1): all three tables are rebuilt by ONE launch on
// stream1, behind the primary rays that counted the hits and ahead of the next ones - a stream that stands idle until
// the next frame's primary rays start, so the 60 us this single workgroup takes on a busy chip delay nobody.  The
// primary rays' table is rewritten in place (only stream1 reads it).  The step-ranked tables are written into the spare
// pair and come into use at the next such dispatch (applyTableSwap), so that neither this frame nor the frames in flight
// wait: the spare pair was in use until the swap before; every frame stream and stream1 are behind everything
// dispatched MAX_FRAMES changes of copies ago (rt_ctx_s::evBefore), so after that many changes nothing can still be
// reading it.  Otherwise - the very first tables, refreshes close together, frames that change no copies, a table
// without a spare, one stream for everything - the frame's stream rebuilds its tables itself, behind the other frames.
void launchOrderBlocks(rt_ctx ctx, hipStream_t st, const rt_ctx_s::BlockMap* m, const OrderTables& tables, int n) {
    unsigned cap = 0;
    for (int i = 0; i < n; i++) { const unsigned r = regionsOf(m, tables.t[i].rs); cap = r > cap ? r : cap; }
    hipLaunchKernelGGL(k_order_blocks, dim3((unsigned)n), dim3(ORDER_THREADS), orderLdsBytes(cap), st, m->bx, m->by, tables, cap);
}
void refreshTables(rt_ctx ctx, rt_ctx_s::BlockMap* m, bool beside) {
    if (!m || !m->dyn) return;
    const bool measured = m->steps && m->stepsValid;
    const bool firstMeasured = measured && !m->rankedBySteps;   // the first frame after measurements exist re-ranks at once
    if (m->dynValid && !firstMeasured && (++m->age % (unsigned)ctx->tableRefreshFrames) != 0) return;
    m->rankedBySteps = measured;
    const uint32_t* const counts = measured ? m->steps : m->cost;
    const int byMax = measured ? 1 : 0;
    const bool oneTable = m->dynRays == m->dyn;
    OrderTables t = {};
    int n = 0;
    // the primary rays' cost follows the hit count, not the secondary rays' steps: their launch has a ranking of its own
    if (m->dynHits) { t.t[n].dyn = m->dynHits; t.t[n].nDyn = m->nDynHits; t.t[n].rs = m->rsHits; t.t[n].costW = m->cost; t.t[n].byMax = 0; n++; }
    const bool aside = beside && m->dynHits && m->dynSpare && m->dynValid && !m->pendingSwap && !ctx->streamShared &&
                       ctx->changes - m->swapChange >= (uint64_t)MAX_FRAMES;
    if (aside) {
        t.t[n].dyn = m->dynSpare; t.t[n].nDyn = m->nDyn; t.t[n].rs = m->rs; t.t[n].costW = counts; t.t[n].byMax = byMax; n++;
        if (!oneTable) { t.t[n].dyn = m->dynRaysSpare; t.t[n].nDyn = m->nDynRays; t.t[n].rs = m->rsRays; t.t[n].costW = counts; t.t[n].byMax = byMax; n++; }
        traceOrderBlocks(ctx, ctx->stream1, m->cost, m->dynHits, nullptr, counts, m->dynSpare, oneTable ? nullptr : m->dynRaysSpare);
        launchOrderBlocks(ctx, ctx->stream1, m, t, n);
        m->pendingSwap = evRecord(ctx, ctx->evTableAside, ctx->stream1) == hipSuccess;
        ctx->asidePending = m->pendingSwap;
        if (!m->pendingSwap) { hipError_t e = hipStreamSynchronize(ctx->stream1); (void)e; applyTableSwap(ctx, m, true); }   // (cannot order it: wait here)
        return;
    }
    if (m->pendingSwap) applyTableSwap(ctx, m);   // (a refresh right after another: first take up what the last one built)
    if (n) {
        hipStream_t st = beside ? ctx->stream1 : SK(ctx);
        traceOrderBlocks(ctx, st, m->cost, m->dynHits);
        launchOrderBlocks(ctx, st, m, t, n);
        if (beside) ctx->asidePending = evRecord(ctx, ctx->evTableAside, st) == hipSuccess || ctx->asidePending;
    }
    // The frame's stream, which waits for the primary rays: behind the kernels of the frames before that may still be
    // walking the tables on the other frame streams, whose next launches wait for the new tables in turn.
    hipStream_t st = SK(ctx);
    if (m->dynValid) {
        joinOthers(ctx);
        if (m->dynSpare) { swapTables(m); m->swapChange = ctx->changes; }
    }
    OrderTables u = {};
    u.t[0].dyn = m->dyn; u.t[0].nDyn = m->nDyn; u.t[0].rs = m->rs; u.t[0].costW = counts; u.t[0].byMax = byMax;
    u.t[1].dyn = m->dynRays; u.t[1].nDyn = m->nDynRays; u.t[1].rs = m->rsRays; u.t[1].costW = counts; u.t[1].byMax = byMax;
    traceOrderBlocks(ctx, st, counts, m->dyn, oneTable ? nullptr : m->dynRays);
    launchOrderBlocks(ctx, st, m, u, oneTable ? 1 : 2);
    if (!ctx->streamShared && evRecord(ctx, ctx->evTable, st) == hipSuccess) {
        for (int k = 0; k < MAX_FRAMES; k++) {
            if (k == ctx->cur || !ctx->frameStream[k]) continue;
            hipError_t e = evWait(ctx, ctx->frameStream[k], ctx->evTable);
            (void)e;
        }
    }
    m->dynValid = true;
}

// Launch bracket: points the kernel at its counter slot and, when timing is on, records events around it.
struct LaunchScope {
    rt_ctx ctx;
    int ik;
    hipEvent_t start = nullptr, stop = nullptr;
    hipStream_t stream;
    LaunchScope(rt_ctx c, int k, KParams* P, hipStream_t on = nullptr, bool merged = false, bool acc = false) : ctx(c), ik(k), stream(on ? on : SK(c)) {
        traceLaunch(ctx, stream, k, P, merged, acc);
        if (ctx->debugDelayUs > 0 && !ctx->noAlloc) {
            ctx->debugDelayState = ctx->debugDelayState * 1664525u + 1013904223u;
            const uint32_t r = ctx->debugDelayState >> 8;
            int bit = stream == ctx->stream1 ? 7 : 0;
            for (int s = 1; s < MAX_FRAMES; s++) if (stream == ctx->frameStream[s]) bit = s;
            if ((r & 1u) && ((ctx->debugDelayMask >> bit) & 1)) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, stream, (unsigned long long)((r >> 1) % (uint32_t)ctx->debugDelayUs) * 100ull);
        }
        if (P) { P->stats = ctx->statsLevel > 0 ? ctx->statsDev + (size_t)k * STAT_SHARDS * 8 : nullptr; P->statsLevel = ctx->statsLevel; }
        ctx->launches[k]++;
        if (!ctx->timing || ctx->noAlloc || (ctx->timingOnly >= 0 && ctx->timingOnly != k)) return;
        auto take = [&]() -> hipEvent_t {
            if (!ctx->freeEvents.empty()) { hipEvent_t e = ctx->freeEvents.back(); ctx->freeEvents.pop_back(); return e; }
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            return e;
        };
        start = take();
        stop = take();
        if (start && stop) { hipError_t e = hipEventRecord(start, stream); (void)e; }
    }
    ~LaunchScope() {
        checkLaunch(ctx, IK_NAMES[ik]);
        // debugging aid (RT_SYNC_LAUNCH=1): wait for every launch, so that a fault is reported with the kernel that made it
        static const bool syncLaunch = [] { const char* e = getenv("RT_SYNC_LAUNCH"); return e && atoi(e) != 0; }();
        if (syncLaunch) {
            const hipError_t e = hipStreamSynchronize(stream);
            if (e != hipSuccess && !ctx->stickyRc) {
                ctx->stickyRc = RT_ERR_LAUNCH;
                ctx->stickyMsg = std::string("kernel '") + IK_NAMES[ik] + "' failed on the device: " + hipGetErrorString(e);
                fprintf(stderr, "rtnative: %s\n", ctx->stickyMsg.c_str());
            }
        }
        if (!start || !stop) return;
        hipError_t e = hipEventRecord(stop, stream);
        (void)e;
        ctx->pending.push_back({ik, start, stop});
    }
};
int drainTiming(rt_ctx ctx) {
    if (ctx->pending.empty()) return RT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
    for (auto& t : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) ctx->timeMs[t.kernel] += ms;
        ctx->freeEvents.push_back(t.start);
        ctx->freeEvents.push_back(t.stop);
    }
    ctx->pending.clear();
    return RT_OK;
}

// Prepares a dispatch whose primary rays run on stream1 beside what the frame streams hold (see rt_ctx_s::stream1,
// frameStream): makes the second copy of every object in `flips`, and of the record buffers in `recs` (Candidates,
// CandidatesB; entries may be null), the current one, orders stream1 and the frame's stream behind the last users of
// those copies, and - `alternate` - moves on to the other frame stream.  Returns false (nothing changed) when that
// cannot be done; the caller then launches on the one dispatch stream as usual.
struct Flip { void** dev; Copies* more; size_t bytes; };
// Makes sure the object has `want` copies in all (new ones are the first to be written: they go to the front of the
// list), zeroed on stream0.  Tagged (record buffers): every copy gets tags; `tagOfCurrent` then points at the tag array
// of the current copy, made here with tag 1 ("older than any frame, newer than the zeros of a new copy").
bool ensureCopies(rt_ctx ctx, Copies& c, size_t bytes, int want, uint32_t** tagOfCurrent, int64_t count) {
    want = want > MAX_FRAMES ? MAX_FRAMES : want;
    const size_t tagBytes = (size_t)count * sizeof(uint32_t);
    if (tagOfCurrent && !*tagOfCurrent && want >= 2) {
        if (devAlloc(ctx, (void**)tagOfCurrent, tagBytes) != hipSuccess) { *tagOfCurrent = nullptr; return false; }
        if (hipMemsetD32Async((hipDeviceptr_t)*tagOfCurrent, 1, (size_t)count, S(ctx)) != hipSuccess) return false;
    }
    while (c.n + 1 < want) {
        void* p = nullptr;
        uint32_t* t = nullptr;
        if (devAlloc(ctx, &p, bytes) != hipSuccess) return false;
        if (tagOfCurrent && devAlloc(ctx, (void**)&t, tagBytes) != hipSuccess) { hipError_t e = devFree(ctx, p); (void)e; return false; }
        // (rows a banded launch never writes must not hold garbage a readback could return)
        bool ok = hipMemsetAsync(p, 0, bytes, S(ctx)) == hipSuccess;
        if (t) ok = ok && hipMemsetAsync(t, 0, tagBytes, S(ctx)) == hipSuccess;
        if (!ok) { hipError_t e = devFree(ctx, p); if (t) e = devFree(ctx, t); (void)e; return false; }
        for (int i = c.n; i > 0; i--) { c.alt[i] = c.alt[i - 1]; c.tag[i] = c.tag[i - 1]; }
        c.alt[0] = p; c.tag[0] = t;
        c.n++;
    }
    return true;
}
// The copy least recently current becomes the current one.
void rotateCopies(void** dev, uint32_t** tag, Copies& c) {
    if (c.n == 0) return;
    void* oldest = c.alt[0];
    uint32_t* oldestTag = c.tag[0];
    for (int i = 0; i + 1 < c.n; i++) { c.alt[i] = c.alt[i + 1]; c.tag[i] = c.tag[i + 1]; }
    c.alt[c.n - 1] = *dev;
    c.tag[c.n - 1] = tag ? *tag : nullptr;
    *dev = oldest;
    if (tag) *tag = oldestTag;
}
bool beginOverlappedPrimary(rt_ctx ctx, const Flip* flips, int n, rt_buffer const* recs, int nRecs, int frames) {
    if (!ctx->stream1 || !ctx->evPrimary) return false;
    const int K = ctx->streamShared || frames < 1 ? 1 : frames > MAX_FRAMES ? MAX_FRAMES : frames;   // frame streams this dispatch alternates between
    const int C = K > 2 ? K : 2;                                              // copies of what the primary rays overwrite
    for (int k = 1; k < K; k++)
        if (!ctx->frameStream[k] && hipStreamCreateWithFlags(&ctx->frameStream[k], hipStreamNonBlocking) != hipSuccess) { ctx->frameStream[k] = nullptr; return false; }
    for (int i = 0; i < n; i++)
        if (!ensureCopies(ctx, *flips[i].more, flips[i].bytes, C, nullptr, 0)) return false;
    for (int i = 0; i < nRecs; i++) {
        rt_buffer b = recs[i];
        if (!b || K < 2) continue;   // one frame stream: the records keep the copies they have
        if (b->layout != LAYOUT_RESERVOIR_RECS && !(b->layout == LAYOUT_AOS && b->stride == 16)) return false;
        const bool fresh = !b->tag;
        if (!ensureCopies(ctx, b->more, layoutBytes(b, b->layout), K, &b->tag, tagCount(b))) {
            if (fresh) dropCopies(b);
            return false;
        }
        if (fresh) ctx->tagged.push_back(b);
    }
    // everything enqueued so far uses (at most) the copies that are current now
    const uint64_t now = ctx->changes;
    const int rec = (int)(now % rt_ctx_s::BEFORE_RING);
    if (evRecord(ctx, ctx->evBefore[rec][0], ctx->stream0) != hipSuccess) return false;
    ctx->beforeValid[rec][0] = true;
    for (int k = 1; k < MAX_FRAMES; k++) {
        ctx->beforeValid[rec][k] = ctx->ahead[k] && evRecord(ctx, ctx->evBefore[rec][k], ctx->frameStream[k]) == hipSuccess;
        if (ctx->ahead[k] && !ctx->beforeValid[rec][k]) return false;
    }
    const int next = K > 1 ? (ctx->cur + 1) % K : 0;
    if (ctx->mainDirty) {  // something else went onto stream0: behind all of it, once
        collapse(ctx);
        if (anyPending(ctx)) return false;
        if (evRecord(ctx, ctx->evTail0, ctx->stream0) != hipSuccess || evWait(ctx, ctx->stream1, ctx->evTail0) != hipSuccess) return false;
        markDirty(ctx);   // (and so do the other frame streams, at their next launch)
        ctx->mainDirty = false;
    } else if (now + 1 >= (uint64_t)C) {
        // the copies written from now on were current until change number now - C + 1 (an object with more copies than
        // C, or one that changes copies less often, was left even earlier)
        const int old = (int)((now + 1 - (uint64_t)C) % rt_ctx_s::BEFORE_RING);
        for (int k = 0; k < MAX_FRAMES; k++) {
            if (!ctx->beforeValid[old][k]) continue;
            if (evWait(ctx, ctx->stream1, ctx->evBefore[old][k]) != hipSuccess) return false;
            if (k != next && evWait(ctx, ctx->frameStream[next], ctx->evBefore[old][k]) != hipSuccess) return false;
        }
    }
    if (ctx->behindCandidates && ctx->afterCandidatesValid && evWait(ctx, ctx->stream1, ctx->afterCandidatesEv) != hipSuccess) return false;
    ctx->changes = now + 1;
    ctx->cur = next;
    ctx->epoch++;
    for (int i = 0; i < n; i++) rotateCopies(flips[i].dev, nullptr, *flips[i].more);
    for (int i = 0; i < nRecs; i++) {
        rt_buffer b = recs[i];
        if (!b || !b->tag) continue;
        rotateCopies(&b->dev, &b->tag, b->more);
        b->epoch = ctx->epoch;
    }
    return true;
}
// After the overlapped launch: the frame's stream goes on only when the primary rays are done.
int endOverlappedPrimary(rt_ctx ctx) {
    HIP_TRY(ctx, evRecord(ctx, ctx->evPrimary, ctx->stream1));
    HIP_TRY(ctx, evWait(ctx, frameStreamNow(ctx), ctx->evPrimary));
    return RT_OK;
}
bool mayOverlap(rt_ctx ctx, bool instr, const KParams& P) {
    if (!ctx->overlap || !ctx->fuse || ctx->noAlloc || instr || ctx->statsLevel >= 2) return false;
    const bool wholeImage = P.row0 <= 0 && P.row1 >= P.ResY;
    return wholeImage || ctx->overlap == 2;
}

// Launch geometry of the per-pixel kernels that share nothing between the waves of a block: by choice
// (rt_ctx_s::oneWaveGroups) one wave per workgroup - four times the workgroups, a quarter of the LDS each - instead of
// the four waves of a 16x16-pixel block.  `kind` = the bit of that choice.  The table has to come in rounds of eight
// entries (every table but the plain row-major one does).
enum { RAYS_SECONDARY = 0, RAYS_PRIMARY = 1, RAYS_RIS_PRIMARY = 2, RAYS_REUSE = 3, RAYS_RIS_CANDIDATES = 4 };
bool oneWaveGroups(rt_ctx ctx, unsigned blocks, int kind) { return ((ctx->oneWaveGroups >> kind) & 1) && blocks % 8 == 0; }
dim3 rayGrid(rt_ctx ctx, unsigned blocks, int kind = RAYS_SECONDARY) { return dim3(oneWaveGroups(ctx, blocks, kind) ? blocks * 4 : blocks); }
dim3 rayBlock(rt_ctx ctx, unsigned blocks, int kind = RAYS_SECONDARY) { return dim3(oneWaveGroups(ctx, blocks, kind) ? 64 : 256); }
size_t rayLds(rt_ctx ctx, unsigned blocks, size_t ldsBytes, int kind = RAYS_SECONDARY) { return oneWaveGroups(ctx, blocks, kind) ? ldsBytes / 4 : ldsBytes; }
// Launches what is held back (rt_ctx_s::Deferred), as it was dispatched.  `acc` = the dispatch that follows is merged into
// it (AccumulateFrames into a held TraceRays, TemporalReuse into a held candidate kernel): the caller has put what that
// dispatch binds into the held parameters.
void launchHeld(rt_ctx ctx, bool acc) {
    rt_ctx_s::Deferred h = ctx->held;
    ctx->held.kind = 0;   // first: everything below may ask for the stream
    // the caller may be an entry point that takes no device of its own (rt_stream, the map calls) in a process that
    // holds contexts on several GPUs
    { hipError_t e = hipSetDevice(ctx->device); (void)e; }
    hipStream_t st = frameStreamNow(ctx);   // (the frame stream cannot have changed since the dispatch was held)
    if (h.kind == 1) {
        LaunchScope ls(ctx, IK_TRANSFER, &h.P, st);
        hipLaunchKernelGGL(k_transfer, dim3(h.blocks), dim3(256), 0, st, h.P);
    } else if (h.kind == 3) {
        if (acc && !h.temporalWaited) waitShared(ctx, rt_ctx_s::SH_TEMPORAL, st);
        {
            size_t topLds = (size_t)rayBlock(ctx, h.blocks).x / 64 * h.P.topCount * 64;   // a copy of the table per wave
            if (rayLds(ctx, h.blocks, h.ldsBytes) + topLds > 65536) { h.P.topCount = 0; topLds = 0; }   // (no room beside the stacks)
            ctx->launchingHeld = true;
            LaunchScope ls(ctx, IK_CANDIDATES, &h.P, st, acc);
            ctx->launchingHeld = false;
            if (h.P.topCount > 0 && acc) hipLaunchKernelGGL((k_candidates<false, true, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes) + topLds, st, h.P);
            else if (h.P.topCount > 0) hipLaunchKernelGGL((k_candidates<false, false, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes) + topLds, st, h.P);
            else if (acc) hipLaunchKernelGGL((k_candidates<false, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes), st, h.P);
            else hipLaunchKernelGGL((k_candidates<false, false>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes), st, h.P);
        }
        if (acc) markShared(ctx, rt_ctx_s::SH_TEMPORAL, st);
        markAfterCandidates(ctx, st, acc && ctx->sharedValid[rt_ctx_s::SH_TEMPORAL][ctx->cur] ? ctx->evShared[rt_ctx_s::SH_TEMPORAL][ctx->cur] : nullptr);
    } else if (h.kind == 4) {
        waitShared(ctx, rt_ctx_s::SH_RESULT, st);
        {
            LaunchScope ls(ctx, IK_RIS_TRACE_RAYS, &h.P, st, false, acc);
            if (acc) hipLaunchKernelGGL((k_ris_trace_rays<false, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes), st, h.P);
            else hipLaunchKernelGGL((k_ris_trace_rays<false, false>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes), st, h.P);
        }
        markShared(ctx, rt_ctx_s::SH_RESULT, st);
    } else if (h.kind == 2) {
        waitShared(ctx, rt_ctx_s::SH_RESULT, st);
        if (h.P.kwWeightRecalc) waitShared(ctx, rt_ctx_s::SH_TEMPORAL, st);   // (RS:645-657 writes into TemporalFrameBuffer)
        {
            size_t topLds = (size_t)rayBlock(ctx, h.blocks).x / 64 * h.P.topCount * 64;
            if (rayLds(ctx, h.blocks, h.ldsBytes) + topLds > 65536) { h.P.topCount = 0; topLds = 0; }
            ctx->launchingHeld = true;
            LaunchScope ls(ctx, IK_TRACE_RAYS, &h.P, st, h.withTransfer, acc);
            ctx->launchingHeld = false;
            if (h.P.topCount > 0 && h.withTransfer && acc) hipLaunchKernelGGL((k_trace_rays<false, true, true, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes) + topLds, st, h.P);
            else if (h.P.topCount > 0 && h.withTransfer) hipLaunchKernelGGL((k_trace_rays<false, true, false, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes) + topLds, st, h.P);
            else if (h.P.topCount > 0 && acc) hipLaunchKernelGGL((k_trace_rays<false, false, true, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes) + topLds, st, h.P);
            else if (h.P.topCount > 0) hipLaunchKernelGGL((k_trace_rays<false, false, false, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes) + topLds, st, h.P);
            else if (h.withTransfer && acc) hipLaunchKernelGGL((k_trace_rays<false, true, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes), st, h.P);
            else if (h.withTransfer) hipLaunchKernelGGL((k_trace_rays<false, true, false>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes), st, h.P);
            else if (acc) hipLaunchKernelGGL((k_trace_rays<false, false, true>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes), st, h.P);
            else hipLaunchKernelGGL((k_trace_rays<false, false, false>), rayGrid(ctx, h.blocks), rayBlock(ctx, h.blocks), rayLds(ctx, h.blocks, h.ldsBytes), st, h.P);
        }
        markShared(ctx, rt_ctx_s::SH_RESULT, st);
        if (h.P.kwWeightRecalc) markShared(ctx, rt_ctx_s::SH_TEMPORAL, st);
    }
}
void launchHeld(rt_ctx ctx) { launchHeld(ctx, false); }
bool mayHold(rt_ctx ctx, bool instr) { return ctx->fuse && !ctx->noAlloc && !instr && ctx->statsLevel < 2; }

int dispatchReSTIR(rt_shader sh, int kernel, int gx, int gy) {
    rt_ctx ctx = sh->ctx;
    KParams P;
    memset(&P, 0, sizeof P);
    fillConstants(sh, P);
    unsigned blocks = 0;
    rt_ctx_s::BlockMap* grid = nullptr;
    int rc = pixelGrid(ctx, P, gx, gy, &blocks, &grid);
    if (rc) return rc;
    const int X = P.ResX, Y = P.ResY;
    const int64_t npx = (int64_t)X * Y;
    auto checkCount = [&](const auto& pl, const char* name) -> int {
        if (pl.count < npx) return fail(ctx, RT_ERR_INVALID_ARG, "buffer '%s' holds %lld records, Resolution needs %lld", name, (long long)pl.count, (long long)npx);
        return RT_OK;
    };
    rt_buffer rec = boundBuffer(sh, kernel, "TraceRecords");
    bool instr = false;
    if (rec) {
        if (rec->stride != RECORD_DWORDS * 4 || rec->count < npx) return fail(ctx, RT_ERR_INVALID_ARG, "TraceRecords must have stride 32 and one record per pixel");
        P.records = (uint32_t*)rec->dev;
        instr = true;
    }
    if (rt_buffer wp = boundBuffer(sh, kernel, "WaveProfile")) {  // diagnostics
        unsigned most = blocks;   // (the longest of the grid's tables: which one a kernel walks is decided below)
        if (grid && grid->dynValid) { most = grid->nDynHits > most ? grid->nDynHits : most; most = grid->nDynRays > most ? grid->nDynRays : most; }
        if (wp->stride != 64 || wp->count < (int64_t)most * 4 + 64) return fail(ctx, RT_ERR_INVALID_ARG, "WaveProfile needs stride 64 and %u records", most * 4 + 64);
        P.waveProfile = (unsigned long long*)wp->dev;
        static const int which = [] { const char* e = getenv("RT_WAVE_PROFILE_KERNEL"); return e ? atoi(e) : 0; }();
        P.waveProfileKernel = which;
        instr = true;
    }
    if (ctx->statsLevel >= 2) instr = true;
    if (ctx->timing && ctx->pending.size() > 8192 && (rc = drainTiming(ctx))) return rc;
    switch (kernel) {
        case 0: {  // InitialTrace
            if ((rc = prepareScene(sh, kernel, P))) return rc;
            rt_buffer lights = boundBuffer(sh, kernel, "LightObjects");
            if (!lights) return fail(ctx, RT_ERR_UNBOUND, "kernel 0: buffer 'LightObjects' is not bound");
            if (lights->stride != 80) return fail(ctx, RT_ERR_INVALID_ARG, "LightObjects stride must be 80");
            P.lights = (const rt_light_object*)lights->dev; P.nLights = lights->count;
            if ((rc = needPlanes(sh, kernel, "HitInfos", LAYOUT_HIT_PLANES, P.hit)) || (rc = checkCount(P.hit, "HitInfos"))) return rc;
            if ((rc = needRecs(sh, kernel, "Candidates", P.cand, false)) || (rc = checkCount(P.cand, "Candidates"))) return rc;
            if ((rc = needTex(sh, kernel, "DebugOverlay", false, X, Y, P.debugOverlay))) return rc;
            if ((rc = needTex(sh, kernel, "DepthBuffer", true, X, Y, P.depth))) return rc;
            if ((rc = needTex(sh, kernel, "NormalsBuffer", false, X, Y, P.normals))) return rc;
            if ((rc = needTex(sh, kernel, "RayHitPointsA", false, X, Y, P.hitA))) return rc;
            if ((rc = needTex(sh, kernel, "RayHitPointsB", false, X, Y, P.hitB))) return rc;
            optionalTex(sh, kernel, "TextureAtlas", P.atlas);
            if (!blocks) break;
            size_t ldsBytes = (size_t)WAVES_PER_BLOCK * P.stackDepth * 64 * 4;
            bool beside = false;
            {
                if (grid) P.blockCost = grid->cost;
                if (grid && grid->dynValid && grid->dynHits) { P.blockMap = grid->dynHits; blocks = grid->nDynHits; }
                // beside the kernels stream0 still holds of the frames before, when nothing forbids it (rt_ctx_s::stream1)
                rt_buffer bHit = boundBuffer(sh, kernel, "HitInfos");
                rt_texture tDepth = boundTexture(sh, kernel, "DepthBuffer");
                rt_texture tPoints = boundTexture(sh, kernel, P.FrameCount % 2 == 0 ? "RayHitPointsA" : "RayHitPointsB");
                SK(ctx);   // (what is held goes first: it was dispatched before this)
                applyTableSwap(ctx, grid);   // tables the last refresh built aside: from this frame on
                const Flip flips[3] = {{&bHit->dev, &bHit->more, layoutBytes(bHit, bHit->layout)}, {&tDepth->dev, &tDepth->more, tDepth->bytes},
                                       {&tPoints->dev, &tPoints->more, tPoints->bytes}};
                // the reservoirs this frame's kernels write: Candidates, and the CandidatesB that SpatialReusePass is bound to
                rt_buffer bCandB = boundBuffer(sh, 1, "CandidatesB");
                if (bCandB && (bCandB->layout != LAYOUT_RESERVOIR_RECS || bCandB->count < npx)) bCandB = nullptr;
                rt_buffer bCand = boundBuffer(sh, kernel, "Candidates");
                rt_buffer const recs[2] = {bCand, bCandB == bCand ? nullptr : bCandB};
                beside = mayOverlap(ctx, instr, P) && beginOverlappedPrimary(ctx, flips, 3, recs, 2, ctx->framesInFlight);
                if (beside) {
                    P.hit = planesOf(bHit);
                    P.depth.r = (float*)tDepth->dev;
                    (P.FrameCount % 2 == 0 ? P.hitA : P.hitB).rgba = (float4*)tPoints->dev;
                    P.cand = recsOf(recs[0]);
                } else {
                    useMainStream(ctx);   // primary rays on the dispatch stream, behind everything
                }
                const bool background = beside && ctx->bgGroups > 0 && ctx->bgCounter && blocks > (unsigned)ctx->bgGroups &&
                                        hipMemsetAsync(ctx->bgCounter, 0, 8 * sizeof(uint32_t), ctx->stream1) == hipSuccess;
                {
                    LaunchScope ls(ctx, IK_PRIMARY, &P, beside ? ctx->stream1 : nullptr);
                    if (background) {
                        P.bgCounter = ctx->bgCounter; P.bgBlocks = blocks;
                        hipLaunchKernelGGL((k_primary<false, true>), dim3((unsigned)ctx->bgGroups), dim3(256), ldsBytes, ls.stream, P);
                    } else if (instr) hipLaunchKernelGGL(k_primary<true>, dim3(blocks), dim3(256), ldsBytes, ls.stream, P);
                    else hipLaunchKernelGGL((k_primary<false>), rayGrid(ctx, blocks, RAYS_PRIMARY), rayBlock(ctx, blocks, RAYS_PRIMARY), rayLds(ctx, blocks, ldsBytes, RAYS_PRIMARY), ls.stream, P);
                }
                if (!beside) markDirty(ctx);   // primary rays on stream0: the next overlapped ones follow them
            }
            // A TemporalReuse merged into the candidate kernel makes it wait for the frames before.  That wait has long
            // been met when the primary rays end, but the stream works its waits off in order, and each costs a few
            // microseconds: it goes ahead of the wait for the primary rays, the one the kernel really stands still for.
            const bool temporalEarly = beside && mayHold(ctx, instr);
            if (temporalEarly) waitShared(ctx, rt_ctx_s::SH_TEMPORAL, frameStreamNow(ctx));
            if (beside && (rc = endOverlappedPrimary(ctx))) return rc;
            refreshTables(ctx, grid, beside);
            if (grid && grid->dyn) {  // the candidate kernel and everything after it run heaviest regions first
                P.blockMap = grid->dynRays;
                blocks = grid->nDynRays;
                P.blockCost = nullptr;
            }
            if (mayHold(ctx, instr)) {  // held: a TemporalReuse that follows becomes part of the launch
                SK(ctx);
                ctx->held.kind = 3; ctx->held.P = P; ctx->held.blocks = blocks; ctx->held.ldsBytes = ldsBytes; ctx->held.grid = grid;
                ctx->held.withTransfer = false; ctx->held.temporalWaited = temporalEarly;
            } else {
                hipStream_t st = SK(ctx);
                {
                    LaunchScope ls(ctx, IK_CANDIDATES, &P, st);
                    if (instr) hipLaunchKernelGGL((k_candidates<true>), dim3(blocks), dim3(256), ldsBytes, st, P);
                    else hipLaunchKernelGGL((k_candidates<false>), rayGrid(ctx, blocks), rayBlock(ctx, blocks), rayLds(ctx, blocks, ldsBytes), st, P);
                }
                markAfterCandidates(ctx, st);
            }
            break;
        }
        case 1: {  // SpatialReusePass
            if ((rc = needRecs(sh, kernel, "Candidates", P.cand)) || (rc = checkCount(P.cand, "Candidates"))) return rc;
            if ((rc = needRecs(sh, kernel, "CandidatesB", P.candB)) || (rc = checkCount(P.candB, "CandidatesB"))) return rc;
            if ((rc = needTex(sh, kernel, "DepthBuffer", true, X, Y, P.depth))) return rc;
            if (!blocks) break;
            {
                // small windows, by choice: the block's neighbourhood staged in LDS (k_spatial_lds), up to the MaxOffset set
                // with RT_SPATIAL_LDS / rt_set_tuning (default 0 = never: at MaxOffset 3 the staged pass takes 50.6 us at
                // 1080p against 38.9 us for the per-lane gathers of k_spatial, whose requests mostly hit in L1 / L2)
                const int ldsMaxOffset = ctx->spatialLdsMaxOffset;
                const int RW = 16 + 2 * P.MaxOffset;
                const size_t regionBytes = (size_t)RW * RW * 68, stageBytes = (size_t)WAVES_PER_BLOCK * 64 * 5 * 16;
                const size_t ldsBytes = regionBytes > stageBytes ? regionBytes : stageBytes;
                bool staged = P.MaxOffset >= 1 && P.MaxOffset <= ldsMaxOffset && ldsBytes <= 160 * 1024;
                if (staged && ldsBytes > 64 * 1024 && !ctx->bigLdsAllowed) {
                    ctx->bigLdsAllowed = hipFuncSetAttribute(reinterpret_cast<const void*>(k_spatial_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
                    staged = ctx->bigLdsAllowed;
                }
                LaunchScope ls(ctx, IK_SPATIAL, &P);
                if (staged) hipLaunchKernelGGL(k_spatial_lds, dim3(blocks), dim3(256), ldsBytes, ls.stream, P);
                else hipLaunchKernelGGL(k_spatial, rayGrid(ctx, blocks, RAYS_REUSE), rayBlock(ctx, blocks, RAYS_REUSE),
                                        rayLds(ctx, blocks, (size_t)WAVES_PER_BLOCK * 64 * COOP_ROW * sizeof(float4), RAYS_REUSE), ls.stream, P);
            }
            break;
        }
        case 2: {  // TransferToOriginal
            if ((rc = needRecs(sh, kernel, "Candidates", P.cand)) || (rc = checkCount(P.cand, "Candidates"))) return rc;
            if ((rc = needRecs(sh, kernel, "CandidatesB", P.candB)) || (rc = checkCount(P.candB, "CandidatesB"))) return rc;
            if ((rc = needTex(sh, kernel, "DepthBuffer", true, X, Y, P.depth))) return rc;
            if (!blocks) break;
            if (mayHold(ctx, instr)) {  // held: a TraceRays that follows takes the copy along
                SK(ctx);
                ctx->held.kind = 1; ctx->held.P = P; ctx->held.blocks = blocks; ctx->held.grid = grid; ctx->held.withTransfer = false;
            } else {
                LaunchScope ls(ctx, IK_TRANSFER, &P);
                hipLaunchKernelGGL(k_transfer, dim3(blocks), dim3(256), 0, ls.stream, P);
            }
            break;
        }
        case 3: {  // TemporalReuse
            if ((rc = needRecs(sh, kernel, "Candidates", P.cand)) || (rc = checkCount(P.cand, "Candidates"))) return rc;
            if ((rc = needPlanes(sh, kernel, "TemporalFrameBuffer", LAYOUT_RESERVOIR_PLANES, P.temporal)) || (rc = checkCount(P.temporal, "TemporalFrameBuffer"))) return rc;
            if ((rc = needTex(sh, kernel, "DepthBuffer", true, X, Y, P.depth))) return rc;
            if ((rc = needTex(sh, kernel, "RayHitPointsA", false, X, Y, P.hitA))) return rc;
            if ((rc = needTex(sh, kernel, "RayHitPointsB", false, X, Y, P.hitB))) return rc;
            if (!blocks) break;
            {
                // a held candidate kernel over the same buffers, rows, table and constants takes the merge along
                rt_ctx_s::Deferred& h = ctx->held;
                const size_t c0 = offsetof(KParams, MaxBVHDepth), c1 = offsetof(KParams, ppFrameCount);
                if (h.kind == 3 && mayHold(ctx, instr) && h.grid == grid && h.P.row0 == P.row0 &&
                    h.P.row1 == P.row1 && h.P.cand.q == P.cand.q && h.P.cand.count == P.cand.count && h.P.depth.r == P.depth.r &&
                    memcmp((const char*)&h.P + c0, (const char*)&P + c0, c1 - c0) == 0) {
                    h.P.temporal = P.temporal; h.P.hitA = P.hitA; h.P.hitB = P.hitB;
                    launchHeld(ctx, true);
                    break;
                }
            }
            {
                hipStream_t st = SK(ctx);
                waitShared(ctx, rt_ctx_s::SH_TEMPORAL, st);
                { LaunchScope ls(ctx, IK_TEMPORAL, &P, st); hipLaunchKernelGGL(k_temporal, dim3(blocks), dim3(256), 0, st, P); }
                markShared(ctx, rt_ctx_s::SH_TEMPORAL, st);
            }
            break;
        }
        case 4: {  // TraceRays
            if ((rc = prepareScene(sh, kernel, P))) return rc;
            if ((rc = needPlanes(sh, kernel, "HitInfos", LAYOUT_HIT_PLANES, P.hit)) || (rc = checkCount(P.hit, "HitInfos"))) return rc;
            if ((rc = needRecs(sh, kernel, "Candidates", P.cand)) || (rc = checkCount(P.cand, "Candidates"))) return rc;
            if ((rc = needPlanes(sh, kernel, "TemporalFrameBuffer", LAYOUT_RESERVOIR_PLANES, P.temporal)) || (rc = checkCount(P.temporal, "TemporalFrameBuffer"))) return rc;
            if ((rc = needTex(sh, kernel, "Result", false, X, Y, P.result))) return rc;
            optionalTex(sh, kernel, "TextureAtlas", P.atlas);
            optionalTex(sh, kernel, "EnvironmentMap", P.env);
            if (!blocks) break;
            size_t ldsBytes = (size_t)WAVES_PER_BLOCK * P.stackDepth * 64 * 4;
            useRaysTable(grid, P, &blocks);
            if (grid && grid->steps) { P.blockSteps = grid->steps; grid->stepsValid = true; }
            if (mayHold(ctx, instr)) {
                // a held TransferToOriginal over the same buffers, rows and table becomes part of this launch
                const rt_ctx_s::Deferred& h = ctx->held;
                const bool takeCopy = h.kind == 1 && h.grid == grid && h.P.cand.q == P.cand.q &&
                                      h.P.cand.count == P.cand.count && h.P.candB.count >= P.cand.count && h.P.row0 == P.row0 && h.P.row1 == P.row1 &&
                                      h.P.ResX == P.ResX && h.P.ResY == P.ResY;
                if (takeCopy) { P.candB = h.P.candB; P.depth = h.P.depth; ctx->held.kind = 0; }
                else SK(ctx);
                ctx->held.kind = 2; ctx->held.P = P; ctx->held.blocks = blocks; ctx->held.ldsBytes = ldsBytes; ctx->held.grid = grid;
                ctx->held.withTransfer = takeCopy;
            } else {
                hipStream_t st = SK(ctx);
                waitShared(ctx, rt_ctx_s::SH_RESULT, st);
                if (P.kwWeightRecalc) waitShared(ctx, rt_ctx_s::SH_TEMPORAL, st);
                {
                    LaunchScope ls(ctx, IK_TRACE_RAYS, &P, st);
                    if (instr) hipLaunchKernelGGL((k_trace_rays<true>), dim3(blocks), dim3(256), ldsBytes, st, P);
                    else hipLaunchKernelGGL((k_trace_rays<false>), rayGrid(ctx, blocks), rayBlock(ctx, blocks), rayLds(ctx, blocks, ldsBytes), st, P);
                }
                markShared(ctx, rt_ctx_s::SH_RESULT, st);
                if (P.kwWeightRecalc) markShared(ctx, rt_ctx_s::SH_TEMPORAL, st);
            }
            break;
        }
        default: return fail(ctx, RT_ERR_INVALID_ARG, "ReSTIR_BVH_RayTracer has no kernel %d", kernel);
    }
    return finish(ctx);
}

// RIS_BVH_RayTracer: GenerateCandidates (0), TraceRays (1).  Bindings as BVH_RIS_DL .../ShaderHelper.cs:7-52 and
// Main.cs:329,352,365 make them.
int dispatchRIS(rt_shader sh, int kernel, int gx, int gy) {
    rt_ctx ctx = sh->ctx;
    KParams P;
    memset(&P, 0, sizeof P);
    fillConstants(sh, P);
    P.RaysPerPixel = (int32_t)cWord(sh, "RaysPerPixel");
    P.SceneObjectCandidatesNum = (int32_t)cWord(sh, "SceneObjectCandidatesNum");
    P.TriCandidatesNum = (int32_t)cWord(sh, "TriCandidatesNum");
    P.TotCandidatesNum = (int32_t)cWord(sh, "TotCandidatesNum");
    P.CandidateReservoirTestsNum = (int32_t)cWord(sh, "CandidateReservoirTestsNum");
    P.risMaxBounceCount = (int32_t)cWord(sh, "MaxBounceCount");
    P.risMaxBVHDepth = (int32_t)cWord(sh, "MaxBVHDepth");
    P.risFrameRand = (int32_t)cWord(sh, "FrameRand");
    P.risSceneObjectReservoirTestsNum = (int32_t)cWord(sh, "SceneObjectReservoirTestsNum");
    P.risTriReservoirTestsNum = (int32_t)cWord(sh, "TriReservoirTestsNum");
    P.risEmittingObjectsNum = (int32_t)cWord(sh, "EmittingObjectsNum");
    unsigned blocks = 0;
    rt_ctx_s::BlockMap* grid = nullptr;
    int rc = pixelGrid(ctx, P, gx, gy, &blocks, &grid);
    if (rc) return rc;
    const int X = P.ResX, Y = P.ResY;
    const int64_t npx = (int64_t)X * Y;
    if (kernel != 0 && kernel != 1) return fail(ctx, RT_ERR_INVALID_ARG, "RIS_BVH_RayTracer has no kernel %d", kernel);
    rt_buffer rec = boundBuffer(sh, kernel, "TraceRecords");
    bool instr = ctx->statsLevel >= 2;
    if (rec) {
        if (rec->stride != RECORD_DWORDS * 4 || rec->count < npx) return fail(ctx, RT_ERR_INVALID_ARG, "TraceRecords must have stride 32 and one record per pixel");
        P.records = (uint32_t*)rec->dev;
        instr = true;
    }
    if (ctx->timing && ctx->pending.size() > 8192 && (rc = drainTiming(ctx))) return rc;
    if ((rc = prepareScene(sh, kernel, P, true))) return rc;
    rt_buffer hit = boundBuffer(sh, kernel, "HitInfos"), cand = boundBuffer(sh, kernel, "Candidates");
    if (!hit || !cand) return fail(ctx, RT_ERR_UNBOUND, "kernel %d needs HitInfos and Candidates bound", kernel);
    if (hit->stride != 40 || cand->stride != 16) return fail(ctx, RT_ERR_INVALID_ARG, "HitInfos stride must be 40 and Candidates 16");
    if (P.RaysPerPixel < 1) return fail(ctx, RT_ERR_INVALID_ARG, "constant 'RaysPerPixel' must be at least 1");
    // every pixel's own record must exist: the candidate kernel reads back what the primary-ray kernel stored
    if (hit->count < (npx - 1) * P.RaysPerPixel + 1)
        return fail(ctx, RT_ERR_INVALID_ARG, "HitInfos holds %lld records, Resolution x RaysPerPixel needs %lld", (long long)hit->count,
                    (long long)((npx - 1) * P.RaysPerPixel + 1));
    if (P.TotCandidatesNum < 0 || P.SceneObjectCandidatesNum < 0 || P.TriCandidatesNum < 0 ||
        (npx * P.RaysPerPixel) * (int64_t)(P.TotCandidatesNum > 0 ? P.TotCandidatesNum : 1) > 0x7fffffffLL)
        return fail(ctx, RT_ERR_LIMIT, "candidate indices would not fit the reference's 32-bit candidate index");
    P.risHit = (rt_ris_hit_info*)hit->dev; P.nRisHit = hit->count;
    P.risCand = (float4*)cand->dev; P.nRisCand = cand->count;
    const size_t ldsBytes = (size_t)WAVES_PER_BLOCK * P.stackDepth * 64 * 4;
    if (kernel == 0) {
        rt_buffer lights = boundBuffer(sh, kernel, "LightObjects");
        if (!lights) return fail(ctx, RT_ERR_UNBOUND, "kernel 0: buffer 'LightObjects' is not bound");
        if (lights->stride != 80) return fail(ctx, RT_ERR_INVALID_ARG, "LightObjects stride must be 80");
        P.lights = (const rt_light_object*)lights->dev; P.nLights = lights->count;
        if ((rc = needTex(sh, kernel, "DebugOverlay", false, X, Y, P.debugOverlay))) return rc;
        if ((rc = needTex(sh, kernel, "RayHitPoints", false, X, Y, P.rayHitPoints))) return rc;
        if (!blocks) return RT_OK;
        bool beside = false;
        {
            if (grid) P.blockCost = grid->cost;
            // beside the kernels the frame streams still hold of the frames before (rt_ctx_s::stream1, frameStream): the RIS
            // project's primary rays overwrite nothing but HitInfos that those still read, and with a copy of Candidates
            // per frame in flight (tagged per pixel ray, like the reservoir records) GenerateCandidates of the next frame
            // runs beside TraceRays of this one; only Result is shared
            SK(ctx);
            applyTableSwap(ctx, grid);   // tables the last refresh built aside: from this frame on
            const int group = P.TotCandidatesNum > 0 ? P.TotCandidatesNum : 1;
            if (cand->tag && cand->tagGroup != group) {   // another number of candidates per ray: the tags no longer fit
                mergeRecords(cand);
                HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
                dropCopies(cand);
            }
            if (!cand->tag) cand->tagGroup = group;
            rt_buffer const recs[1] = {cand->count % group == 0 && cand->layout == LAYOUT_AOS ? cand : nullptr};
            const Flip flips[1] = {{&hit->dev, &hit->more, hit->bytes}};
            beside = mayOverlap(ctx, instr, P) && beginOverlappedPrimary(ctx, flips, 1, recs, 1, ctx->risFramesInFlight);
            if (!beside) useMainStream(ctx);
            P.risCand = (float4*)cand->dev;
            P.risTag = cand->tag; P.risTagValue = (uint32_t)ctx->epoch;
            if (beside) P.risHit = (rt_ris_hit_info*)hit->dev;
            {
                LaunchScope ls(ctx, IK_RIS_PRIMARY, &P, beside ? ctx->stream1 : nullptr);
                if (instr) hipLaunchKernelGGL(k_ris_primary<true>, dim3(blocks), dim3(256), ldsBytes, ls.stream, P);
                else hipLaunchKernelGGL((k_ris_primary<false>), rayGrid(ctx, blocks, RAYS_RIS_PRIMARY), rayBlock(ctx, blocks, RAYS_RIS_PRIMARY), rayLds(ctx, blocks, ldsBytes, RAYS_RIS_PRIMARY), ls.stream, P);
            }
            if (!beside) markDirty(ctx);   // primary rays on stream0: the next overlapped ones follow them
        }
        if (beside && (rc = endOverlappedPrimary(ctx))) return rc;
        refreshTables(ctx, grid, beside);
        if (grid && grid->dyn) {
            P.blockMap = grid->dyn;
            blocks = grid->nDyn;
            P.blockCost = nullptr;
        }
        if (cand->epoch != ctx->epoch) joinOthers(ctx);   // Candidates without a copy for this frame: behind the frame before
        {
            LaunchScope ls(ctx, IK_RIS_CANDIDATES, &P);
            hipLaunchKernelGGL(k_ris_candidates, rayGrid(ctx, blocks, RAYS_RIS_CANDIDATES), rayBlock(ctx, blocks, RAYS_RIS_CANDIDATES),
                               rayLds(ctx, blocks, WAVES_PER_BLOCK * risStageBytes, RAYS_RIS_CANDIDATES), ls.stream, P);
            // (no markAfterCandidates: the RIS project's primary rays do best beside the whole of the frame before,
            // 0.756 against 0.784 ms per 1080p frame)
        }
    } else {
        if ((rc = needTex(sh, kernel, "Result", false, X, Y, P.result))) return rc;
        if (!blocks) return RT_OK;
        useRaysTable(grid, P, &blocks);
        if (cand->epoch != ctx->epoch) joinOthers(ctx);
        if (mayHold(ctx, instr)) {  // held: an AccumulateFrames that follows becomes part of the launch
            SK(ctx);
            ctx->held.kind = 4; ctx->held.P = P; ctx->held.blocks = blocks; ctx->held.ldsBytes = ldsBytes; ctx->held.grid = grid;
            ctx->held.withTransfer = false;
        } else {
            hipStream_t st = SK(ctx);
            waitShared(ctx, rt_ctx_s::SH_RESULT, st);
            {
                LaunchScope ls(ctx, IK_RIS_TRACE_RAYS, &P, st);
                if (instr) hipLaunchKernelGGL((k_ris_trace_rays<true>), dim3(blocks), dim3(256), ldsBytes, st, P);
                else hipLaunchKernelGGL((k_ris_trace_rays<false>), rayGrid(ctx, blocks), rayBlock(ctx, blocks), rayLds(ctx, blocks, ldsBytes), st, P);
            }
            markShared(ctx, rt_ctx_s::SH_RESULT, st);
        }
    }
    return finish(ctx);
}

int dispatchPreCalc(rt_shader sh, int kernel, int gx) {
    rt_ctx ctx = sh->ctx;
    if (kernel != 0) return fail(ctx, RT_ERR_INVALID_ARG, "PreCalc has no kernel %d", kernel);
    // Both projects name their pre-calculation shader "PreCalc"; the RIS project's binds "Tris" + "SceneObjects" and
    // sets "TrisNum" (BVH_RIS_DL .../ShaderHelper.cs:16-18,31), the ReSTIR project's "Triangles" + "Vertices".
    if (rt_buffer rTris = boundBuffer(sh, 0, "Tris")) {
        rt_buffer rObjs = boundBuffer(sh, 0, "SceneObjects");
        if (!rObjs) return fail(ctx, RT_ERR_UNBOUND, "CalcTriNormals needs Tris and SceneObjects bound");
        if (rTris->stride != 76 || rObjs->stride != 168) return fail(ctx, RT_ERR_INVALID_ARG, "CalcTriNormals: Tris stride must be 76 and SceneObjects 168");
        uint32_t trisNum = cWord(sh, "TrisNum");
        if (gx <= 0) return RT_OK;
        uint64_t covered = (uint64_t)gx * 256;
        uint32_t n = trisNum < covered ? trisNum : (uint32_t)covered;
        {
            LaunchScope ls(ctx, IK_RIS_CALC_NORMALS, nullptr);
            hipLaunchKernelGGL(k_ris_calc_tri_normals, dim3((unsigned)gx), dim3(256), 0, S(ctx), (rt_ris_tri*)rTris->dev, rTris->count,
                               (const rt_scene_object*)rObjs->dev, rObjs->count, n);
        }
        rTris->version = ctx->versionCounter++;
        return finish(ctx);
    }
    rt_buffer bTris = boundBuffer(sh, 0, "Triangles"), bVerts = boundBuffer(sh, 0, "Vertices");
    if (!bTris || !bVerts) return fail(ctx, RT_ERR_UNBOUND, "CalcTriNormals needs Triangles and Vertices bound");
    if (bTris->stride != 28 || bVerts->stride != 20) return fail(ctx, RT_ERR_INVALID_ARG, "CalcTriNormals: Triangles stride must be 28 and Vertices 20");
    uint32_t trianglesNum = cWord(sh, "TrianglesNum");
    if (gx <= 0) return RT_OK;
    // threads beyond the dispatched groups do not run, exactly as with the reference's dispatch
    uint64_t covered = (uint64_t)gx * 256;
    uint32_t n = trianglesNum < covered ? trianglesNum : (uint32_t)covered;
    {
        LaunchScope ls(ctx, IK_CALC_NORMALS, nullptr);
        hipLaunchKernelGGL(k_calc_tri_normals, dim3((unsigned)gx), dim3(256), 0, S(ctx), (rt_triangle*)bTris->dev, bTris->count,
                           (const rt_vertex*)bVerts->dev, bVerts->count, n);
    }
    bTris->version = ctx->versionCounter++;
    return finish(ctx);
}

int dispatchPostProc(rt_shader sh, int kernel, int gx, int gy) {
    rt_ctx ctx = sh->ctx;
    if (kernel != 0) return fail(ctx, RT_ERR_INVALID_ARG, "PostProcessor has no kernel %d", kernel);
    KParams P;
    memset(&P, 0, sizeof P);
    rt_texture res = boundTexture(sh, 0, "Result"), acc = boundTexture(sh, 0, "AccumulatedResult");
    if (!res || !acc) return fail(ctx, RT_ERR_UNBOUND, "AccumulateFrames needs Result and AccumulatedResult bound");
    if (res->format == RT_FORMAT_R32F || acc->format == RT_FORMAT_R32F || res->w != acc->w || res->h != acc->h)
        return fail(ctx, RT_ERR_INVALID_ARG, "AccumulateFrames: Result and AccumulatedResult must be RGBA textures of equal size");
    // The kernel has no bounds check (PostProcessor.compute:11-18); threads outside the textures read 0 and
    // their writes are dropped, so the covered region is the intersection of the dispatch and the texture.
    P.ResX = res->w; P.ResY = res->h;
    P.ppFrameCount = (int32_t)cWord(sh, "FrameCount");
    P.result.rgba = (float4*)res->dev; P.accumulated.rgba = (float4*)acc->dev;
    P.accumulated.unorm8 = acc->format == RT_FORMAT_RGBA8_UNORM;
    unsigned blocks = 0;
    rt_ctx_s::BlockMap* grid = nullptr;
    int rc = pixelGrid(ctx, P, gx, gy, &blocks, &grid);
    if (rc) return rc;
    {
        // a held TraceRays that writes this Result over the same rows and table takes the running average along
        rt_ctx_s::Deferred& h = ctx->held;
        if (blocks && (h.kind == 2 || h.kind == 4) && ctx->fuse && !ctx->noAlloc && ctx->statsLevel < 2 && h.grid == grid &&
            h.P.result.rgba == P.result.rgba && h.P.ResX == P.ResX && h.P.ResY == P.ResY && h.P.row0 == P.row0 && h.P.row1 == P.row1) {
            h.P.accumulated = P.accumulated;
            h.P.ppFrameCount = P.ppFrameCount;
            launchHeld(ctx, true);
            return finish(ctx);
        }
    }
    if (blocks) {
        hipStream_t st = SK(ctx);
        waitShared(ctx, rt_ctx_s::SH_RESULT, st);
        { LaunchScope ls(ctx, IK_ACCUMULATE, &P, st); hipLaunchKernelGGL(k_accumulate, dim3(blocks), dim3(256), 0, st, P); }
        markShared(ctx, rt_ctx_s::SH_RESULT, st);
    }
    return finish(ctx);
}

}  // namespace

// =============================================================== C-ABI ===============================================================
extern "C" {

int rt_create(int device_id, rt_ctx* out) {
    if (!out) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return RT_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return RT_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RT_ERR_NO_DEVICE;  // the only code object in this library
    rt_ctx ctx = new rt_ctx_s;
    ctx->device = device_id;
    static const bool fuse = [] { const char* e = getenv("RT_FUSE"); return e ? atoi(e) != 0 : true; }();
    ctx->fuse = fuse;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream0, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->noAllocDone, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void**)&ctx->statsDev, IK_COUNT * STAT_SHARDS * 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemsetAsync(ctx->statsDev, 0, IK_COUNT * STAT_SHARDS * 8 * sizeof(unsigned long long), S(ctx)) != hipSuccess) {
        delete ctx;
        return RT_ERR_NO_DEVICE;
    }
    {   // the stream of the overlapped primary rays.  Default priority: a lowest-priority stream measured no better
        // (0.752 against 0.745 ms per 1080p frame), and in a process in which RCCL had created a communicator a wait
        // of stream0 on an event of a lowest-priority stream never returned (RT_OVERLAP_PRIORITY=1 selects it anyway).
        static const int overlap = [] { const char* e = getenv("RT_OVERLAP"); return e ? atoi(e) : 1; }();
        ctx->overlap = overlap < 0 || overlap > 2 ? 1 : overlap;
        static const bool behindCand = [] { const char* e = getenv("RT_OVERLAP_START"); return e ? atoi(e) != 0 : true; }();
        ctx->behindCandidates = behindCand;
        static const int spatialLds = [] { const char* e = getenv("RT_SPATIAL_LDS"); return e ? atoi(e) : 0; }();
        ctx->spatialLdsMaxOffset = spatialLds;
        static const int bg = [] { const char* e = getenv("RT_PRIMARY_BG"); return e ? atoi(e) : 0; }();
        ctx->bgGroups = bg < 0 ? 0 : (bg + 7) / 8 * 8;
        if (hipMalloc((void**)&ctx->bgCounter, 8 * sizeof(uint32_t)) != hipSuccess) ctx->bgCounter = nullptr;
        int least = 0, greatest = 0;
        static const bool lowPriority = [] { const char* e = getenv("RT_OVERLAP_PRIORITY"); return e ? atoi(e) != 0 : false; }();
        if (!lowPriority || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        static const int inFlight = [] { const char* e = getenv("RT_FRAMES_IN_FLIGHT"); return e ? atoi(e) : 2; }();
        ctx->framesInFlight = inFlight < 1 ? 1 : inFlight > MAX_FRAMES ? MAX_FRAMES : inFlight;
        static const int risInFlight = [] { const char* e = getenv("RT_RIS_FRAMES_IN_FLIGHT"); return e ? atoi(e) : 3; }();
        ctx->risFramesInFlight = risInFlight < 1 ? 1 : risInFlight > MAX_FRAMES ? MAX_FRAMES : risInFlight;
        ctx->frameStream[0] = ctx->stream0;
        bool ok = hipStreamCreateWithPriority(&ctx->stream1, hipStreamNonBlocking, least) == hipSuccess;
        // (the third and fourth dispatch stream are made when they are asked for: a process has few hardware queues -
        // four by default - and streams that share one run one after the other)
        for (int k = 1; k < ctx->framesInFlight; k++) ok = ok && hipStreamCreateWithFlags(&ctx->frameStream[k], hipStreamNonBlocking) == hipSuccess;
        std::vector<hipEvent_t*> evs = {&ctx->evAfterCandidates, &ctx->evPrimary, &ctx->evTail0, &ctx->evTable, &ctx->evTableAside};
        for (auto& ev : ctx->evJoin) evs.push_back(&ev);
        for (auto& ev : ctx->evOther) evs.push_back(&ev);
        for (auto& per : ctx->evShared) for (auto& ev : per) evs.push_back(&ev);
        for (auto& per : ctx->evBefore) for (auto& ev : per) evs.push_back(&ev);
        for (hipEvent_t* ev : evs) ok = ok && hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            rt_destroy(ctx);
            return RT_ERR_NO_DEVICE;
        }
    }
    {   // released device memory kept for reuse: a sixteenth of the device's memory at most (RT_DEVICE_POOL_MB overrides, 0 = none)
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) totalB = (size_t)64 << 30;
        static const long poolMb = [] { const char* e = getenv("RT_DEVICE_POOL_MB"); return e ? atol(e) : -1L; }();
        ctx->pool.capBytes = poolMb >= 0 ? (size_t)poolMb << 20 : totalB / 16;
    }
    // this device's copy of the file-scope scratch array (every GPU has its own, at its own address)
    if (hipGetSymbolAddress((void**)&ctx->scratch, HIP_SYMBOL(scratch)) != hipSuccess || !ctx->scratch) {
        rt_destroy(ctx);
        return RT_ERR_NO_DEVICE;
    }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx ctx) {
    if (!ctx) return;
    hipError_t e = hipSetDevice(ctx->device);
    if (ctx->held.kind) launchHeld(ctx);
    for (int k = 1; k < MAX_FRAMES; k++) if (ctx->frameStream[k]) e = hipStreamSynchronize(ctx->frameStream[k]);
    e = hipStreamSynchronize(ctx->stream0);
    {
        std::lock_guard<std::mutex> lock(scratchMutex);
        for (auto& o : scratchOwner) if (o == ctx) o = nullptr;
    }
    for (auto& pr : ctx->preps) {
        if (pr.wide) e = devFree(ctx, pr.wide);
        if (pr.btri) e = devFree(ctx, pr.btri);
        if (pr.objRoot) e = devFree(ctx, pr.objRoot);
        if (pr.top) e = devFree(ctx, pr.top);
        if (pr.objRootTop) e = devFree(ctx, pr.objRootTop);
    }
    ctx->pool.capBytes = 0;
    devPoolTrim(ctx, 0);
    if (ctx->statsDev) e = hipFree(ctx->statsDev);
    if (ctx->bgCounter) e = hipFree(ctx->bgCounter);
    for (rt_shader sh : ctx->shaders) sh->ctx = nullptr;   // a shader closed after its context only frees itself
    if (ctx->noAllocDone) e = hipEventDestroy(ctx->noAllocDone);
    freeBlockMaps(ctx);
    for (auto& t : ctx->pending) { e = hipEventDestroy(t.start); e = hipEventDestroy(t.stop); }
    for (auto ev : ctx->freeEvents) e = hipEventDestroy(ev);
    for (auto ev : ctx->orderEvent) if (ev) e = hipEventDestroy(ev);
    for (auto ev : ctx->marks) if (ev) e = hipEventDestroy(ev);
    if (ctx->transferStream) { e = hipStreamSynchronize(ctx->transferStream); e = hipStreamDestroy(ctx->transferStream); }
    if (ctx->stream1) { e = hipStreamSynchronize(ctx->stream1); e = hipStreamDestroy(ctx->stream1); }
    for (int k = 1; k < MAX_FRAMES; k++) if (ctx->frameStream[k]) { e = hipStreamSynchronize(ctx->frameStream[k]); e = hipStreamDestroy(ctx->frameStream[k]); }
    for (hipEvent_t ev : {ctx->evPrimary, ctx->evAfterCandidates, ctx->evTail0, ctx->evTable, ctx->evTableAside}) if (ev) e = hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->evJoin) if (ev) e = hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->evOther) if (ev) e = hipEventDestroy(ev);
    for (auto& per : ctx->evShared) for (hipEvent_t ev : per) if (ev) e = hipEventDestroy(ev);
    for (auto& per : ctx->evBefore) for (hipEvent_t ev : per) if (ev) e = hipEventDestroy(ev);
    e = hipStreamDestroy(ctx->stream0);
    (void)e;
    delete ctx;
}

int rt_sync(rt_ctx ctx) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    // (the overlapped primary rays are included: the frame streams wait for them)
    SK(ctx);
    for (int k = 1; k < MAX_FRAMES; k++) {
        if (!ctx->ahead[k]) continue;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->frameStream[k]));
        ctx->ahead[k] = false;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream0));
    for (int k = 1; k < MAX_FRAMES; k++) ctx->dirty[k] = false;   // nothing is left anywhere that a later launch would have to wait for
    if (ctx->transferStream) HIP_TRY(ctx, hipStreamSynchronize(ctx->transferStream));
    if (ctx->tracing) ctx->traceOps.push_back({'Y', 0, nullptr, "sync", 0, {}});   // the host has waited for every stream
    return finish(ctx);
}

void* rt_transfer_stream(rt_ctx ctx) {
    if (!ctx) return nullptr;
    if (!ctx->transferStream) {
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamCreateWithFlags(&ctx->transferStream, hipStreamNonBlocking) != hipSuccess) {
            ctx->transferStream = nullptr;
            fail(ctx, RT_ERR_NO_DEVICE, "the transfer stream could not be created");
        }
    }
    return (void*)ctx->transferStream;
}
int rt_stream_order(rt_ctx ctx, int first) {
    if (!ctx || first < 0 || first > 1) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    if (!rt_transfer_stream(ctx)) return RT_ERR_NO_DEVICE;
    hipEvent_t& ev = ctx->orderEvent[first];
    if (!ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipStream_t from = first == 0 ? SK(ctx) : ctx->transferStream, to = first == 0 ? ctx->transferStream : SK(ctx);
    HIP_TRY(ctx, evRecord(ctx, ev, from));
    HIP_TRY(ctx, evWait(ctx, to, ev));
    return finish(ctx);
}
int rt_mark(rt_ctx ctx, int stream, int slot) {
    if (!ctx || stream < 0 || stream > 1 || slot < 0 || slot >= 64) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    if (stream == 1 && !rt_transfer_stream(ctx)) return RT_ERR_NO_DEVICE;
    if (!ctx->marks[slot]) HIP_TRY(ctx, hipEventCreate(&ctx->marks[slot]));
    HIP_TRY(ctx, evRecord(ctx, ctx->marks[slot], stream == 0 ? SK(ctx) : ctx->transferStream));
    return finish(ctx);
}
int rt_wait_mark(rt_ctx ctx, int stream, int slot) {
    if (!ctx || stream < 0 || stream > 1 || slot < 0 || slot >= 64) return RT_ERR_INVALID_ARG;
    if (!ctx->marks[slot]) return fail(ctx, RT_ERR_INVALID_ARG, "stamp %d was never recorded", slot);
    int rc = useDevice(ctx);
    if (rc) return rc;
    if (stream == 1 && !rt_transfer_stream(ctx)) return RT_ERR_NO_DEVICE;
    HIP_TRY(ctx, evWait(ctx, stream == 0 ? SK(ctx) : ctx->transferStream, ctx->marks[slot]));
    return finish(ctx);
}
int rt_mark_elapsed(rt_ctx ctx, int a, int b, double* ms) {
    if (!ctx || !ms || a < 0 || a >= 64 || b < 0 || b >= 64) return RT_ERR_INVALID_ARG;
    if (!ctx->marks[a] || !ctx->marks[b]) return fail(ctx, RT_ERR_INVALID_ARG, "stamp %d or %d was never recorded", a, b);
    int rc = useDevice(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventSynchronize(ctx->marks[a]));
    HIP_TRY(ctx, hipEventSynchronize(ctx->marks[b]));
    float f = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&f, ctx->marks[a], ctx->marks[b]));
    *ms = f;
    return finish(ctx);
}
const char* rt_last_error(rt_ctx ctx) { return ctx ? ctx->err.c_str() : "null context"; }
void* rt_stream(rt_ctx ctx) {
    if (!ctx) return nullptr;
    // the host will put work of its own on the stream: from now on every dispatch of this context goes there too
    useMainStream(ctx);
    ctx->streamShared = true;
    return (void*)S(ctx);
}
int rt_flush(rt_ctx ctx) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    SK(ctx);
    return finish(ctx);
}
int rt_set_overlap(rt_ctx ctx, int mode) {
    if (!ctx || mode < 0 || mode > 2) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    SK(ctx);
    static const bool processOff = [] { const char* e = getenv("RT_OVERLAP"); return e && atoi(e) == 0; }();
    ctx->overlap = processOff ? 0 : mode;   // RT_OVERLAP=0 switches it off for the whole process, whatever is asked for later
    ctx->afterCandidatesValid = false;
    markDirty(ctx);   // the next overlapped primary rays follow all of stream0
    return finish(ctx);
}
int rt_set_tuning(rt_ctx ctx, const char* name, int value) {
    if (!ctx || !name) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    SK(ctx);
    if (!strcmp(name, "spatial_lds_max_offset")) ctx->spatialLdsMaxOffset = value < 0 ? 0 : value;
    else if (!strcmp(name, "primary_background_groups")) ctx->bgGroups = value < 0 ? 0 : (value + 7) / 8 * 8;
    else if (!strcmp(name, "overlap_behind_candidates")) { ctx->behindCandidates = value != 0; ctx->afterCandidatesValid = false; markDirty(ctx); }
    else if (!strcmp(name, "debug_stream_delay_us")) ctx->debugDelayUs = value < 0 ? 0 : value > 2000 ? 2000 : value;
    else if (!strcmp(name, "debug_stream_delay_mask")) ctx->debugDelayMask = value & 0xff;
    else if (!strcmp(name, "region_blocks_reuse") || !strcmp(name, "region_blocks_primary") || !strcmp(name, "region_blocks_rays")) {
        if (value != 1 && value != 2 && value != 4 && value != 8) return fail(ctx, RT_ERR_INVALID_ARG, "%s must be 1, 2, 4 or 8", name);
        // the tables are built when a grid is first used: forget them once nothing can still be reading them
        useMainStream(ctx);
        HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
        freeBlockMaps(ctx);
        ctx->regionBlocks[!strcmp(name, "region_blocks_reuse") ? 0 : !strcmp(name, "region_blocks_primary") ? 1 : 2] = value;
    }
    else if (!strcmp(name, "top_nodes_lds")) {   // the prepared scenes are rebuilt with or without the table at their next use
        if (value < 0 || value > TOP_MAX) return fail(ctx, RT_ERR_INVALID_ARG, "top_nodes_lds must be 0 to %d", TOP_MAX);
        useMainStream(ctx);
        HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
        ctx->topNodes = value;
        for (auto& pr : ctx->preps) pr.valid = false;
    }
    else if (!strcmp(name, "reuse_validation")) { ctx->reuseValidation = value != 0; if (!value) { ctx->validated[0] = rt_ctx_s::Validated(); ctx->validated[1] = rt_ctx_s::Validated(); } }
    else if (!strcmp(name, "device_pool_mb")) {   // released device memory kept for reuse at most; what is kept beyond it is freed now
        if (value < 0) return fail(ctx, RT_ERR_INVALID_ARG, "device_pool_mb must not be negative");
        ctx->pool.capBytes = (size_t)value << 20;
        devPoolTrim(ctx, ctx->pool.capBytes);
    }
    else if (!strcmp(name, "one_wave_groups")) {
        if (value < 0 || value > 31) return fail(ctx, RT_ERR_INVALID_ARG, "one_wave_groups is a mask of bits 0 to 4");
        ctx->oneWaveGroups = value;
    }
    else if (!strcmp(name, "table_refresh_frames")) {
        if (value < 1) return fail(ctx, RT_ERR_INVALID_ARG, "table_refresh_frames must be at least 1");
        ctx->tableRefreshFrames = value;
    }
    else if (!strcmp(name, "ris_frames_in_flight")) {
        if (value < 1 || value > MAX_FRAMES) return fail(ctx, RT_ERR_INVALID_ARG, "ris_frames_in_flight must be 1 to %d", MAX_FRAMES);
        useMainStream(ctx);
        ctx->risFramesInFlight = value;
        if (value >= 2) ctx->streamShared = false;
        markDirty(ctx);
    }
    else if (!strcmp(name, "frames_in_flight")) {
        if (value < 1 || value > MAX_FRAMES) return fail(ctx, RT_ERR_INVALID_ARG, "frames_in_flight must be 1 to %d", MAX_FRAMES);
        useMainStream(ctx);
        for (int k = 1; k < value; k++)
            if (!ctx->frameStream[k]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->frameStream[k], hipStreamNonBlocking));
        ctx->framesInFlight = value;
        if (value >= 2) ctx->streamShared = false;   // (the host's word that it no longer uses the stream rt_stream gave it)
        markDirty(ctx);
    }
    else return fail(ctx, RT_ERR_INVALID_ARG, "unknown tuning parameter '%s'", name);
    return finish(ctx);
}
int rt_debug_trace(rt_ctx ctx, int on) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    SK(ctx);
    if (on) ctx->traceOps.clear();
    ctx->tracing = on != 0;
    return finish(ctx);
}
int rt_debug_note_rows(rt_ctx ctx, int stream, rt_buffer buf, rt_texture tex, int row0, int row1, int write) {
    if (!ctx || stream < 0 || stream > 1 || (!buf && !tex) || row0 < 0 || row1 < row0) return RT_ERR_INVALID_ARG;
    if ((buf && buf->ctx != ctx) || (tex && tex->ctx != ctx)) return RT_ERR_INVALID_ARG;
    if (!ctx->tracing) return RT_OK;
    TraceLaunch t(ctx, stream == 0 ? frameStreamNow(ctx) : ctx->transferStream, "transfer");
    if (stream == 1) t.op.stream = MAX_FRAMES + 1;
    if (buf) t.add(buf->dev, write != 0, false, row0, row1);
    if (tex) t.add(tex->dev, write != 0, false, row0, row1);
    ctx->traceOps.push_back(t.op);
    return RT_OK;
}
int64_t rt_debug_trace_dump(rt_ctx ctx, void* out, int64_t capacity) {
    if (!ctx || capacity < 0 || (!out && capacity)) return RT_ERR_INVALID_ARG;
    std::string text;
    char line[128];
    for (const auto& op : ctx->traceOps) {
        if (op.kind == 'L') {
            snprintf(line, sizeof line, "L %d %s", op.stream, op.name);
            text += line;
            for (int i = 0; i < op.nAcc; i++) {
                snprintf(line, sizeof line, " %p:%c%s", op.acc[i].obj, op.acc[i].write ? 'w' : 'r', op.acc[i].relaxed ? "~" : "");
                text += line;
                if (op.acc[i].row1 >= 0) { snprintf(line, sizeof line, ":%d-%d", op.acc[i].row0, op.acc[i].row1); text += line; }
            }
            text += "\n";
        } else if (op.kind == 'H') {
            text += "H 0\n";
        } else if (op.kind == 'Y') {
            text += "Y 0\n";
        } else {
            snprintf(line, sizeof line, "%c %d %p\n", op.kind, op.stream, op.event);
            text += line;
        }
    }
    const int64_t need = (int64_t)text.size() + 1;
    if (out && capacity > 0) {
        const int64_t n = need <= capacity ? need - 1 : capacity - 1;
        memcpy(out, text.data(), (size_t)n);
        ((char*)out)[n] = 0;
    }
    return need;
}
int rt_set_fusion(rt_ctx ctx, int enabled) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    S(ctx);
    ctx->fuse = enabled != 0;
    return finish(ctx);
}

int rt_set_row_band(rt_ctx ctx, int row0, int row1) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (row1 > 0 && (row0 < 0 || row0 > row1)) return fail(ctx, RT_ERR_INVALID_ARG, "row band [%d,%d) is not valid", row0, row1);
    ctx->row0 = row0; ctx->row1 = row1;
    return RT_OK;
}

int rt_shader_open(rt_ctx ctx, const char* name, rt_shader* out) {
    if (!ctx || !name || !out) return RT_ERR_INVALID_ARG;
    int kind;
    if (!strcmp(name, "ReSTIR_BVH_RayTracer")) kind = SH_RESTIR;
    else if (!strcmp(name, "RIS_BVH_RayTracer")) kind = SH_RIS;
    else if (!strcmp(name, "PreCalc")) kind = SH_PRECALC;
    else if (!strcmp(name, "PostProcessor") || !strcmp(name, "PostProcesser")) kind = SH_POSTPROC;  // the RIS project's file is PostProcesser.compute
    else return fail(ctx, RT_ERR_INVALID_ARG, "unknown shader '%s'", name);
    rt_shader sh = new rt_shader_s;
    sh->ctx = ctx;
    sh->kind = kind;
    ctx->shaders.push_back(sh);
    *out = sh;
    return RT_OK;
}
void rt_shader_close(rt_shader sh) {
    if (!sh) return;
    if (sh->ctx) {
        auto& v = sh->ctx->shaders;
        v.erase(std::remove(v.begin(), v.end(), sh), v.end());
    }
    delete sh;
}

int rt_find_kernel(rt_shader sh, const char* name) {
    if (!sh || !name) return RT_ERR_INVALID_ARG;
    static const char* restir[] = {"InitialTrace", "SpatialReusePass", "TransferToOriginal", "TemporalReuse", "TraceRays"};
    switch (sh->kind) {
        case SH_RESTIR:
            for (int i = 0; i < 5; i++) if (!strcmp(name, restir[i])) return i;
            break;
        case SH_RIS:
            if (!strcmp(name, "GenerateCandidates")) return 0;
            if (!strcmp(name, "TraceRays")) return 1;
            break;
        case SH_PRECALC: if (!strcmp(name, "CalcTriNormals")) return 0; break;
        case SH_POSTPROC: if (!strcmp(name, "AccumulateFrames")) return 0; break;
    }
    return fail(sh->ctx, RT_ERR_INVALID_ARG, "kernel '%s' not found", name);
}

int rt_dispatch(rt_shader sh, int kernel, int gx, int gy, int gz) {
    if (!sh) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = sh->ctx;
    if (kernel < 0 || kernel >= 8) return fail(ctx, RT_ERR_INVALID_ARG, "kernel index %d out of range", kernel);
    if (gx < 0 || gy < 0 || gz < 0) return fail(ctx, RT_ERR_INVALID_ARG, "negative thread-group count");
    int rc = useDevice(ctx);
    if (rc) return rc;
    if (gx == 0 || gy == 0 || gz == 0) return finish(ctx);
    switch (sh->kind) {
        case SH_RESTIR: return dispatchReSTIR(sh, kernel, gx, gy);
        case SH_RIS: return dispatchRIS(sh, kernel, gx, gy);
        case SH_PRECALC: return dispatchPreCalc(sh, kernel, gx);
        case SH_POSTPROC: return dispatchPostProc(sh, kernel, gx, gy);
    }
    return fail(ctx, RT_ERR_INVALID_ARG, "shader kind %d cannot be dispatched", sh->kind);
}

int rt_dispatch_no_alloc(rt_shader sh, int kernel, int gx, int gy, int gz) {
    if (!sh) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = sh->ctx;
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (ctx->device >= 0 && ctx->device < SCRATCH_MAX_DEVICES) {
        // every context of this GPU shares one scratch array: refuse the call while another context's launches may
        // still be reading the table it left there (hipEventQuery allocates nothing)
        std::lock_guard<std::mutex> lock(scratchMutex);
        rt_ctx_s*& owner = scratchOwner[ctx->device];
        if (owner && owner != ctx && hipEventQuery(owner->noAllocDone) == hipErrorNotReady)
            return fail(ctx, RT_ERR_BUSY, "rt_dispatch_no_alloc: another context's dispatch is still using this GPU's scratch array");
        owner = ctx;
    }
    ctx->noAlloc = true;  // every path below that would allocate (or free) device memory returns an error instead
    { hipError_t e = hipSetDevice(ctx->device); (void)e; }
    useMainStream(ctx);   // (enqueues two stream operations at most; allocates nothing)
    int rc = rt_dispatch(sh, kernel, gx, gy, gz);
    ctx->noAlloc = false;
    { hipError_t e = evRecord(ctx, ctx->noAllocDone, ctx->stream0); (void)e; }
    return rc;
}

int rt_enable_keyword(rt_shader sh, const char* kw) {
    if (!sh || !kw) return RT_ERR_INVALID_ARG;
    if (!strcmp(kw, "VIS_REUSE")) sh->kwVisReuse = true;
    else if (!strcmp(kw, "WEIGHT_RECALC")) sh->kwWeightRecalc = true;
    else if (!strcmp(kw, "BRDF")) sh->kwBrdf = true;
    return RT_OK;
}
int rt_disable_keyword(rt_shader sh, const char* kw) {
    if (!sh || !kw) return RT_ERR_INVALID_ARG;
    if (!strcmp(kw, "VIS_REUSE")) sh->kwVisReuse = false;
    else if (!strcmp(kw, "WEIGHT_RECALC")) sh->kwWeightRecalc = false;
    else if (!strcmp(kw, "BRDF")) sh->kwBrdf = false;
    return RT_OK;
}

static int setWords(rt_shader sh, const char* name, const void* src, int nwords) {
    if (!sh || !name || (!src && nwords)) return RT_ERR_INVALID_ARG;
    if (nwords < 0 || nwords > 64) return fail(sh->ctx, RT_ERR_INVALID_ARG, "constant '%s': %d words", name, nwords);
    std::vector<uint32_t>& v = sh->constants[name];
    v.resize((size_t)nwords);
    if (nwords) memcpy(v.data(), src, (size_t)nwords * 4);
    return RT_OK;
}
int rt_set_int(rt_shader sh, const char* name, int value) { return setWords(sh, name, &value, 1); }
int rt_set_float(rt_shader sh, const char* name, float value) { return setWords(sh, name, &value, 1); }
int rt_set_bool(rt_shader sh, const char* name, int value) { int v = value ? 1 : 0; return setWords(sh, name, &v, 1); }
int rt_set_ints(rt_shader sh, const char* name, const int* values, int count) { return setWords(sh, name, values, count); }
int rt_set_vector(rt_shader sh, const char* name, const float values[4]) { return setWords(sh, name, values, 4); }
int rt_set_matrix(rt_shader sh, const char* name, const float values[16]) { return setWords(sh, name, values, 16); }

int rt_set_buffer(rt_shader sh, int kernel, const char* name, rt_buffer buf) {
    if (!sh || !name) return RT_ERR_INVALID_ARG;
    if (kernel < 0 || kernel >= 8) return fail(sh->ctx, RT_ERR_INVALID_ARG, "kernel index %d out of range", kernel);
    if (buf && buf->ctx != sh->ctx) return fail(sh->ctx, RT_ERR_INVALID_ARG, "buffer belongs to another context");
    if (buf) sh->buffers[kernel][name] = buf; else sh->buffers[kernel].erase(name);
    return RT_OK;
}
int rt_set_texture(rt_shader sh, int kernel, const char* name, rt_texture tex) {
    if (!sh || !name) return RT_ERR_INVALID_ARG;
    if (kernel < 0 || kernel >= 8) return fail(sh->ctx, RT_ERR_INVALID_ARG, "kernel index %d out of range", kernel);
    if (tex && tex->ctx != sh->ctx) return fail(sh->ctx, RT_ERR_INVALID_ARG, "texture belongs to another context");
    if (tex) sh->textures[kernel][name] = tex; else sh->textures[kernel].erase(name);
    return RT_OK;
}

int rt_buffer_create(rt_ctx ctx, int64_t count, int stride, rt_buffer* out) {
    if (!ctx || !out) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    if (count < 1 || stride < 4 || stride % 4 != 0) return fail(ctx, RT_ERR_INVALID_ARG, "buffer count %lld / stride %d invalid", (long long)count, stride);
    int rc = useDevice(ctx);
    if (rc) return rc;
    rt_buffer b = new rt_buffer_s;
    b->ctx = ctx; b->count = count; b->stride = stride; b->bytes = (size_t)count * stride; b->layout = LAYOUT_AOS;
    b->version = ctx->versionCounter++;
    b->dev = nullptr;
    hipError_t e = devAlloc(ctx, &b->dev, b->bytes);
    if (e == hipSuccess) e = hipMemsetAsync(b->dev, 0, b->bytes, S(ctx));
    b->zeroed = e == hipSuccess;
    if (e != hipSuccess) {
        if (b->dev) { hipError_t e2 = devFree(ctx, b->dev); (void)e2; }
        delete b;
        return fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_NO_DEVICE, "buffer allocation of %zu bytes failed: %s", (size_t)count * stride, hipGetErrorString(e));
    }
    *out = b;
    return finish(ctx);
}

int rt_buffer_set_data(rt_buffer b, const void* src, size_t bytes) {
    if (!b || (!src && bytes)) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = b->ctx;
    if (bytes > b->bytes) return fail(ctx, RT_ERR_INVALID_ARG, "set_data of %zu bytes into a %zu-byte buffer", bytes, b->bytes);
    int rc = useDevice(ctx);
    if (rc) return rc;
    if (b->more.n && !b->tag) settleRecords(ctx);
    if (b->tag && b->layout == LAYOUT_AOS) {   // copies of a buffer that is kept in the ABI layout: back to one
        mergeRecords(b);
        HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
        dropCopies(b);
    }
    if ((rc = convertLayout(b, LAYOUT_AOS))) return rc;
    b->zeroed = false;
    // synchronous copy: the caller may reuse `src` as soon as this returns
    HIP_TRY(ctx, hipMemcpyAsync(b->dev, src, bytes, hipMemcpyHostToDevice, S(ctx)));
    HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
    b->version = ctx->versionCounter++;
    return finish(ctx);
}

int rt_buffer_get_data(rt_buffer b, void* dst, size_t bytes) {
    if (!b || (!dst && bytes)) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = b->ctx;
    if (bytes > b->bytes) return fail(ctx, RT_ERR_INVALID_ARG, "get_data of %zu bytes from a %zu-byte buffer", bytes, b->bytes);
    int rc = useDevice(ctx);
    if (rc) return rc;
    if (b->layout == LAYOUT_AOS) {
        mergeRecords(b);   // (the RIS project's Candidates with a copy per frame in flight: the current copy made complete)
        HIP_TRY(ctx, hipMemcpyAsync(dst, b->dev, bytes, hipMemcpyDeviceToHost, S(ctx)));
        HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
        return finish(ctx);
    }
    void* tmp = nullptr;
    HIP_TRY(ctx, devAlloc(ctx, &tmp, b->bytes));
    launchToAos(b, tmp);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(dst, tmp, bytes, hipMemcpyDeviceToHost, S(ctx));
    if (e == hipSuccess) e = hipStreamSynchronize(S(ctx));
    hipError_t e2 = devFree(ctx, tmp);
    (void)e2;
    if (e != hipSuccess) return fail(ctx, RT_ERR_NO_DEVICE, "buffer readback failed: %s", hipGetErrorString(e));
    return finish(ctx);
}

int rt_buffer_get_range(rt_buffer b, void* dst, int64_t first, int64_t count) {
    if (!b || (!dst && count)) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = b->ctx;
    if (first < 0 || count < 0 || first + count > b->count)
        return fail(ctx, RT_ERR_INVALID_ARG, "elements [%lld, %lld) are outside a buffer of %lld", (long long)first, (long long)(first + count), (long long)b->count);
    if (count == 0) return finish(ctx);
    int rc = useDevice(ctx);
    if (rc) return rc;
    const size_t bytes = (size_t)count * b->stride;
    if (b->layout == LAYOUT_AOS) {
        mergeRecords(b);
        HIP_TRY(ctx, hipMemcpyAsync(dst, (const char*)b->dev + (size_t)first * b->stride, bytes, hipMemcpyDeviceToHost, S(ctx)));
        HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
        return finish(ctx);
    }
    void* tmp = nullptr;
    HIP_TRY(ctx, devAlloc(ctx, &tmp, bytes));
    launchToAos(b, tmp, first, count);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(dst, tmp, bytes, hipMemcpyDeviceToHost, S(ctx));
    if (e == hipSuccess) e = hipStreamSynchronize(S(ctx));
    hipError_t e2 = devFree(ctx, tmp);
    (void)e2;
    if (e != hipSuccess) return fail(ctx, RT_ERR_NO_DEVICE, "buffer readback failed: %s", hipGetErrorString(e));
    return finish(ctx);
}

// Unity's Release leaves the ComputeBuffer object unusable; a shader that still has it bound fails its next dispatch.
// Here the handle is taken out of every binding table of the context's shaders, so that such a dispatch fails with
// RT_ERR_UNBOUND instead of reading freed memory, and the traversal arrays derived from a scene buffer go with it.
void rt_buffer_release(rt_buffer b) {
    if (!b) return;
    rt_ctx ctx = b->ctx;
    hipError_t e = hipSetDevice(ctx->device);
    e = hipStreamSynchronize(S(ctx));   // launches what is held first: it may read this buffer
    if (ctx->transferStream) e = hipStreamSynchronize(ctx->transferStream);
    for (rt_shader sh : ctx->shaders)
        for (auto& table : sh->buffers)
            for (auto it = table.begin(); it != table.end();) it = it->second == b ? table.erase(it) : std::next(it);
    for (auto& pr : ctx->preps) {
        if (pr.pBVs != b->dev && pr.pTris != b->dev && pr.pVerts != b->dev && pr.pObjs != b->dev) continue;
        if (pr.wide) e = devFree(ctx, pr.wide);
        if (pr.btri) e = devFree(ctx, pr.btri);
        if (pr.objRoot) e = devFree(ctx, pr.objRoot);
        if (pr.top) e = devFree(ctx, pr.top);
        if (pr.objRootTop) e = devFree(ctx, pr.objRootTop);
        pr = rt_ctx_s::Prep();
    }
    if (b->tag) dropCopies(b);
    freeCopies(ctx, b->more);
    if (b->dev) e = devFree(ctx, b->dev);
    (void)e;
    delete b;
}
int64_t rt_buffer_count(rt_buffer b) { return b ? b->count : 0; }
int rt_buffer_stride(rt_buffer b) { return b ? b->stride : 0; }

int rt_texture_create(rt_ctx ctx, int w, int h, rt_format format, rt_texture* out) {
    if (!ctx || !out) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    if (w < 1 || h < 1 || w > 65536 || h > 65536) return fail(ctx, RT_ERR_INVALID_ARG, "texture size %dx%d invalid", w, h);
    if (format != RT_FORMAT_R32F && format != RT_FORMAT_RGBA32F && format != RT_FORMAT_RGBA8_UNORM) return fail(ctx, RT_ERR_INVALID_ARG, "unknown texture format %d", (int)format);
    int rc = useDevice(ctx);
    if (rc) return rc;
    rt_texture t = new rt_texture_s;
    t->ctx = ctx; t->w = w; t->h = h; t->format = format;
    t->version = ctx->versionCounter++;
    t->bytes = (size_t)w * h * (format == RT_FORMAT_R32F ? 4 : 16);
    t->dev = nullptr;
    hipError_t e = devAlloc(ctx, &t->dev, t->bytes);
    if (e == hipSuccess) e = hipMemsetAsync(t->dev, 0, t->bytes, S(ctx));
    if (e != hipSuccess) {
        if (t->dev) { hipError_t e2 = devFree(ctx, t->dev); (void)e2; }
        delete t;
        return fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_NO_DEVICE, "texture allocation failed: %s", hipGetErrorString(e));
    }
    *out = t;
    return finish(ctx);
}

int rt_texture_set_data(rt_texture t, const void* src, size_t bytes) {
    if (!t || !src) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = t->ctx;
    int rc = useDevice(ctx);
    if (rc) return rc;
    const int64_t n = (int64_t)t->w * t->h;
    if (t->format == RT_FORMAT_RGBA8_UNORM) {
        if (bytes != (size_t)n * 4) return fail(ctx, RT_ERR_INVALID_ARG, "RGBA8 texture upload must be %lld bytes", (long long)n * 4);
        void* tmp = nullptr;
        HIP_TRY(ctx, devAlloc(ctx, &tmp, bytes));
        hipError_t e = hipMemcpyAsync(tmp, src, bytes, hipMemcpyHostToDevice, S(ctx));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_rgba8_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(ctx), (const uint8_t*)tmp, (float4*)t->dev, n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(S(ctx));
        hipError_t e2 = devFree(ctx, tmp);
        (void)e2;
        if (e != hipSuccess) return fail(ctx, RT_ERR_NO_DEVICE, "texture upload failed: %s", hipGetErrorString(e));
        return finish(ctx);
    }
    if (bytes != t->bytes) return fail(ctx, RT_ERR_INVALID_ARG, "texture upload must be %zu bytes", t->bytes);
    if (t->more.n) settleRecords(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(t->dev, src, bytes, hipMemcpyHostToDevice, S(ctx)));
    HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
    t->version = ctx->versionCounter++;
    return finish(ctx);
}

int rt_texture_get_data(rt_texture t, void* dst, size_t bytes) {
    if (!t || !dst) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = t->ctx;
    if (bytes > t->bytes) return fail(ctx, RT_ERR_INVALID_ARG, "texture readback of %zu bytes from %zu", bytes, t->bytes);
    int rc = useDevice(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(dst, t->dev, bytes, hipMemcpyDeviceToHost, S(ctx)));
    HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
    return finish(ctx);
}

int rt_texture_get_rows(rt_texture t, void* dst, int row0, int row1) {
    if (!t || !dst) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = t->ctx;
    if (row0 < 0 || row1 < row0 || row1 > t->h) return fail(ctx, RT_ERR_INVALID_ARG, "rows [%d,%d) are outside the texture", row0, row1);
    if (row1 == row0) return finish(ctx);
    int rc = useDevice(ctx);
    if (rc) return rc;
    const size_t rowBytes = (size_t)t->w * (t->format == RT_FORMAT_R32F ? 4 : 16);
    HIP_TRY(ctx, hipMemcpyAsync(dst, (const char*)t->dev + rowBytes * row0, rowBytes * (size_t)(row1 - row0), hipMemcpyDeviceToHost, S(ctx)));
    HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
    return finish(ctx);
}

void rt_texture_release(rt_texture t) {
    if (!t) return;
    rt_ctx ctx = t->ctx;
    hipError_t e = hipSetDevice(ctx->device);
    e = hipStreamSynchronize(S(ctx));
    if (ctx->transferStream) e = hipStreamSynchronize(ctx->transferStream);
    for (rt_shader sh : ctx->shaders)   // as rt_buffer_release: no binding may keep the handle
        for (auto& table : sh->textures)
            for (auto it = table.begin(); it != table.end();) it = it->second == t ? table.erase(it) : std::next(it);
    if (t->dev) e = devFree(ctx, t->dev);
    freeCopies(ctx, t->more);
    (void)e;
    delete t;
}

int rt_buffer_map_rows(rt_buffer b, int width, int row0, int row1, rt_segment out[4], int* count) {
    if (!b || !out || !count) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = b->ctx;
    if (width < 1 || row0 < 0 || row1 < row0 || (int64_t)row1 * width > b->count)
        return fail(ctx, RT_ERR_INVALID_ARG, "rows [%d,%d) of width %d are outside the buffer", row0, row1, width);
    SK(ctx);   // whoever reads through these pointers orders itself against the stream: nothing may still be held back
    b->zeroed = false;   // (the host may write through them)
    const size_t first = (size_t)row0 * width, n = (size_t)(row1 - row0) * width;
    // An object with two copies: the address is that of the current copy.  Unless the context renders row bands beside
    // one another (overlap 2, where the rows a band does not own are undefined in such objects anyway), that copy is
    // first made to hold what a single buffer would.
    if (ctx->overlap != 2) {
        if (b->tag && b->layout == LAYOUT_RESERVOIR_RECS) mergeRecords(b, (int64_t)first, (int64_t)n);
        else if (b->tag) mergeRecords(b);
        else if (b->more.n) settleRecords(ctx);
    }
    char* base = (char*)b->dev;
    if (b->layout == LAYOUT_AOS) {
        out[0].ptr = base + first * b->stride; out[0].bytes = n * b->stride;
        *count = 1;
    } else if (b->layout == LAYOUT_RESERVOIR_RECS) {
        out[0].ptr = base + first * 64; out[0].bytes = n * 64;
        *count = 1;
    } else {
        for (int k = 0; k < 3; k++) { out[k].ptr = base + (size_t)b->count * 16 * k + first * 16; out[k].bytes = n * 16; }
        out[3].ptr = base + (size_t)b->count * 48 + first * 4; out[3].bytes = n * 4;
        *count = 4;
    }
    return finish(ctx);
}
int rt_buffer_rows_written(rt_buffer b, int width, int row0, int row1) {
    if (!b) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = b->ctx;
    if (width < 1 || row0 < 0 || row1 < row0 || (int64_t)row1 * width > b->count)
        return fail(ctx, RT_ERR_INVALID_ARG, "rows [%d,%d) of width %d are outside the buffer", row0, row1, width);
    int rc = useDevice(ctx);
    if (rc) return rc;
    if (!b->tag || row1 == row0) return finish(ctx);
    // (only the frames that work on this copy write its tags, and they do so on this stream or behind it)
    const size_t per = b->layout == LAYOUT_RESERVOIR_RECS ? 1 : (size_t)b->tagGroup;   // records per tag
    const size_t t0 = (size_t)row0 * width / per, t1 = ((size_t)row1 * width + per - 1) / per;
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)(b->tag + t0), (int)(uint32_t)ctx->epoch, t1 - t0, SK(ctx)));
    return finish(ctx);
}
int rt_texture_map_rows(rt_texture t, int row0, int row1, rt_segment* out) {
    if (!t || !out) return RT_ERR_INVALID_ARG;
    if (row0 < 0 || row1 < row0 || row1 > t->h) return fail(t->ctx, RT_ERR_INVALID_ARG, "rows [%d,%d) are outside the texture", row0, row1);
    SK(t->ctx);
    if (t->more.n && t->ctx->overlap != 2) settleRecords(t->ctx);
    const size_t texel = t->format == RT_FORMAT_R32F ? 4 : 16;
    out->ptr = (char*)t->dev + (size_t)row0 * t->w * texel;
    out->bytes = (size_t)(row1 - row0) * t->w * texel;
    return finish(t->ctx);
}

int rt_enable_stats(rt_ctx ctx, int level) {
    if (!ctx || level < 0 || level > 2) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    if (level != ctx->statsLevel)  // counters restart whenever the level changes
        HIP_TRY(ctx, hipMemsetAsync(ctx->statsDev, 0, IK_COUNT * STAT_SHARDS * 8 * sizeof(unsigned long long), S(ctx)));
    ctx->statsLevel = level;
    return finish(ctx);
}
static int fetchStats(rt_ctx ctx, int ik, bool reset, unsigned long long h[8]) {
    unsigned long long raw[STAT_SHARDS * 8];
    unsigned long long* src = ctx->statsDev + (size_t)ik * STAT_SHARDS * 8;
    HIP_TRY(ctx, hipMemcpyAsync(raw, src, sizeof raw, hipMemcpyDeviceToHost, S(ctx)));
    if (reset) HIP_TRY(ctx, hipMemsetAsync(src, 0, sizeof raw, S(ctx)));
    HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
    for (int c = 0; c < 8; c++) { h[c] = 0; for (int s = 0; s < STAT_SHARDS; s++) h[c] += raw[s * 8 + c]; }
    return finish(ctx);
}
static int readStats(rt_ctx ctx, int ik, rt_dispatch_stats* acc) {
    unsigned long long h[8];
    int rc = fetchStats(ctx, ik, true, h);
    if (rc) return rc;
    acc->rays += h[ST_RAYS]; acc->pops += h[ST_POPS]; acc->bvChecks += h[ST_BVCHECKS]; acc->objVisits += h[ST_OBJVISITS];
    acc->triIters += h[ST_TRIITERS]; acc->triChecks += h[ST_TRICHECKS]; acc->validNeighbours += h[ST_NEIGHBOURS];
    acc->activePixels += h[ST_ACTIVE];
    return finish(ctx);
}
int rt_get_stats(rt_shader sh, int kernel, rt_dispatch_stats* out) {
    if (!sh || !out || kernel < 0 || kernel >= 8) return RT_ERR_INVALID_ARG;
    rt_ctx ctx = sh->ctx;
    int rc = useDevice(ctx);
    if (rc) return rc;
    memset(out, 0, sizeof *out);
    if (sh->kind == SH_RESTIR) {
        static const int map[5][2] = {{IK_PRIMARY, IK_CANDIDATES}, {IK_SPATIAL, -1}, {IK_TRANSFER, -1}, {IK_TEMPORAL, -1}, {IK_TRACE_RAYS, -1}};
        if (kernel >= 5) return fail(ctx, RT_ERR_INVALID_ARG, "no kernel %d", kernel);
        for (int j = 0; j < 2; j++)
            if (map[kernel][j] >= 0 && (rc = readStats(ctx, map[kernel][j], out))) return rc;
        return finish(ctx);
    }
    if (sh->kind == SH_RIS) {
        if (kernel == 0) { if ((rc = readStats(ctx, IK_RIS_PRIMARY, out))) return rc; return readStats(ctx, IK_RIS_CANDIDATES, out); }
        if (kernel == 1) return readStats(ctx, IK_RIS_TRACE_RAYS, out);
        return fail(ctx, RT_ERR_INVALID_ARG, "no kernel %d", kernel);
    }
    if (sh->kind == SH_POSTPROC && kernel == 0) return readStats(ctx, IK_ACCUMULATE, out);
    return fail(ctx, RT_ERR_INVALID_ARG, "no statistics are kept for this shader's kernel %d", kernel);
}

int rt_profile_enable(rt_ctx ctx, int timing) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    if ((rc = drainTiming(ctx))) return rc;
    ctx->timing = timing != 0;
    return finish(ctx);
}
int rt_profile_select(rt_ctx ctx, const char* kernel_name) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    int ik = -1;
    if (kernel_name && *kernel_name) {
        for (int i = 0; i < IK_COUNT; i++) if (!strcmp(kernel_name, IK_NAMES[i])) ik = i;
        if (ik < 0) return fail(ctx, RT_ERR_INVALID_ARG, "unknown kernel name '%s'", kernel_name);
    }
    ctx->timingOnly = ik;
    return finish(ctx);
}
int rt_profile_reset(rt_ctx ctx) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    if ((rc = drainTiming(ctx))) return rc;
    for (int i = 0; i < IK_COUNT; i++) { ctx->timeMs[i] = 0; ctx->launches[i] = 0; }
    HIP_TRY(ctx, hipMemsetAsync(ctx->statsDev, 0, IK_COUNT * STAT_SHARDS * 8 * sizeof(unsigned long long), S(ctx)));
    HIP_TRY(ctx, hipStreamSynchronize(S(ctx)));
    return finish(ctx);
}
int rt_profile_get(rt_ctx ctx, const char* kernel_name, rt_kernel_profile* out) {
    if (!ctx || !kernel_name || !out) return RT_ERR_INVALID_ARG;
    int rc = useDevice(ctx);
    if (rc) return rc;
    int ik = -1;
    for (int i = 0; i < IK_COUNT; i++) if (!strcmp(kernel_name, IK_NAMES[i])) ik = i;
    if (ik < 0) return fail(ctx, RT_ERR_INVALID_ARG, "unknown kernel name '%s'", kernel_name);
    if ((rc = drainTiming(ctx))) return rc;
    memset(out, 0, sizeof *out);
    out->launches = ctx->launches[ik];
    out->milliseconds = ctx->timeMs[ik];
    unsigned long long h[8];
    if ((rc = fetchStats(ctx, ik, false, h))) return rc;
    out->stats.rays = h[ST_RAYS]; out->stats.pops = h[ST_POPS]; out->stats.bvChecks = h[ST_BVCHECKS];
    out->stats.objVisits = h[ST_OBJVISITS]; out->stats.triIters = h[ST_TRIITERS]; out->stats.triChecks = h[ST_TRICHECKS];
    out->stats.validNeighbours = h[ST_NEIGHBOURS]; out->stats.activePixels = h[ST_ACTIVE];
    return finish(ctx);
}

}  // extern "C"
