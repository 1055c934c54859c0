// libfyprt.so — C ABI (include/fyprt.h) over the gfx950 kernels of rt_kernels.h.
// Owns all device memory of one renderer context; everything stays resident in HBM between
// frames (the reference re-allocates and round-trips ~100 MB over PCIe per 1080p frame,
// Renderer.cu:37-53, :70, :244-283 — SURVEY.md §8 a14).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "rt_host.h"
#include <hipcub/hipcub.hpp>
#include "rt_lbvh.h"
#include "rt_refit.h"
#include "rt_paths.h"

using namespace rt;

namespace {

thread_local std::string g_createError;

bool g_hostOnlyAlloc = false;   // set while a host-only context ingests a scene (no device allocations)
template <class T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    hipError_t alloc(size_t count) {
        release(); n = count;
        if (count == 0 || g_hostOnlyAlloc) return hipSuccess;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    size_t bytes() const { return n * sizeof(T); }
};

// host mat4 product, same operation order as the device / glm (column j = ((a0*bj.x + a1*bj.y) + a2*bj.z) + a3*bj.w)
void matmul_cm(const float* a, const float* b, float* out) {
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) {
            float t = a[0 * 4 + r] * b[j * 4 + 0] + a[1 * 4 + r] * b[j * 4 + 1];
            t = t + a[2 * 4 + r] * b[j * 4 + 2];
            t = t + a[3 * 4 + r] * b[j * 4 + 3];
            out[j * 4 + r] = t;
        }
}

}  // namespace

struct ncclUniqueIdBytes { char b[128]; };        // == ncclUniqueId (rccl.h: 128 opaque bytes, passed by value)

struct fyprt_context {
    // multi-process multi-GPU (fyprt_comm_*, fyprt_multi.h): RCCL communicator of the ranks that share the frame, every rank's band
    void* comm = nullptr; int world = 1, rank = 0; std::vector<uint32_t> bounds; int commHaloMode = 0;
    int device = 0; hipStream_t stream = nullptr; std::string err; bool hostOnly = false;
    // ReSTIR DI frames are pipelined over two streams: Part 1 + Part-2 setup of frame N+1 (front stream) run beside the
    // persistent trace kernel of frame N (`stream`, on which every frame COMPLETES and which fyprt_stream() hands out)
    hipStream_t front = nullptr; hipEvent_t evFront[2] = {}, evDone[2] = {}; bool lastOverlapped = false; bool ringSplit[128] = {};
    static constexpr int kRing = 128;          // frames whose per-launch hipEvents are kept (fyprt_frame_timings)
    hipEvent_t ring[kRing][5] = {}; int ringLaunches[kRing] = {}; unsigned long long frameSerial = 0; hipEvent_t* ev = nullptr;
    uint32_t W = 0, H = 0, frameIndex = 1, rowBegin = 0, rowEnd = 0, halo = 0; bool rowsSet = false;
    uint32_t stripeRows = 0, stripeParts = 1, stripePart = 0;       // fyprt_set_row_stripes (per-pixel techniques only)
    uint32_t commStripeRows = 0, commLastStripeRows = 0; bool commLastStriped = false;
    int lastBuildRounds = 0;                                          // PLOC rounds of the last device build (diagnostic)
    bool haloExchange = false;   // halo rows of ReSTIR Part 1 come from the bands that own them (fyprt_multi.h) instead of being recomputed here
    bool part1Pending = false;   // fyprt_render_part(1) was called, part 2 must follow
    bool blockingCall = false;   // inside fyprt_render (which synchronises anyway): long path stages may poll the live-path count from the host
    uint32_t histDI[2] = {0, 0}, histGI[2] = {0, 0};   // rows [begin, end) whose ReSTIR DI / GI history this context holds (the band of the last such frame)
    bool haveScene = false, haveCamera = false, countRays = false;
    // per-pixel buffers
    DevBuf<float4> accum; DevBuf<uint32_t> image; DevBuf<Payload> payload; DevBuf<float> depth; DevBuf<f2> normalA, normalB;
    DevBuf<DIRes> di, diPrev; DevBuf<GIRes> gi, giPrev; DevBuf<float4> giHot; bool normalFlip = false;
    DevBuf<DIRec> drec, dprevA, dprevB; bool dprevFlip = false; int lastTech = -1;
    int lastRestir = -1;                                              // technique of the last ReSTIR frame (-1: none since the buffers were zeroed): k_sync_history_normals
    uint32_t* externalImage = nullptr;
    // scene
    // what a device refit needs beyond the tree itself (fyprt_update_vertices): vertices, per-triangle vertex indices, the nodes of
    // every level (bottom level first), a box per node; and the topology the host light-tree builder is fed again
    DevBuf<DevVertex> dverts; DevBuf<uint4> triIdx; DevBuf<uint32_t> levelNodes; DevBuf<float4> nodeBox; std::vector<uint32_t> levelOffset;
    // fyprt_update_transforms: object-space vertices on the device, the meshes' vertex ranges, a host copy of the world vertices (the host
    // light-tree builder reads the emissive meshes' vertices from it)
    DevBuf<DevVertex> objVerts; std::vector<uint32_t> meshFirstVertex; std::vector<fyprt_vertex> hostVerts;
    std::vector<uint32_t> topoTris; std::vector<fyprt_mesh> topoMeshes; std::vector<fyprt_material> topoMats; uint32_t vertexCount = 0; bool prebuiltLightTrees = false;
    bool hostBvhStale = false;                      // the device tree was refitted: fyprt_export_bvh reads it back first
    bool hostVertsStale = false;                    // world vertices were recomputed on the device: the host copy only follows for emissive meshes
    DevBuf<float4> nodes, leafTris, triPos, triShade, mats; DevBuf<DevTexture> texTable; std::vector<DevBuf<uint32_t>> texPixels;
    DevBuf<uint32_t> emissive; DevBuf<float4> lightRecs; DevBuf<DevLTNode> ltTlas, ltBlas; DevBuf<uint32_t> ltFirst, ltCount, ltRoot, ltLeafOfTri;
    DevBuf<unsigned long long> rayCounter;
    DevScene dsc{}; DevCamera dcam{};
    rth::SceneBVH hostBvh; rth::LightTrees hostLt; uint32_t meshCount = 0;
    int lastLaunches = 0;
    size_t queueStride = 0;                     // float4s per task queue
    size_t sortGroups = 0;                      // setup workgroups the sort scratch is sized for (per parity)
    int traceOcc = 0; size_t traceOccLds = 0;   // cached residency of the persistent trace kernel
    int tuning[24] = {2, 1, 0, 0, 128, 24, 24, /*7: node-loop quorum of the primary-ray kernels*/ 32, 0, /*9: static chunks, 0 = auto*/ 0, 32, 1, 0, 0, 0, 0, /*16: top nodes kept in LDS*/ 0, /*17: fused small-scene frame*/ 0, /*18: skip dead shadow rays*/ 1, /*19: ReSTIR GI Part 2 in one launch*/ 2, /*20: its service threshold*/ 48, 0, 0, 0};   // [0] tile order  [1] DI part 2: 0 one thread per pixel, 1 wavefront queue + persistent trace  [2] persistent workgroups per CU
    int numCUs = 256;
    // wavefront path engine (rt_paths.h): two ray lists + results (ping-pong), per-pixel path state, pixel lists, list counters
    DevBuf<float4> wfRays[2], wfHits[2], wfState; DevBuf<uint32_t> wfPixels, wfPixels2, wfCounters;
    int pathOcc = 0; size_t pathOccLds = 0;     // cached residency of k_trace_rays
    int gi2Occ = 0; size_t gi2OccLds = 0;       // ... of k_gi2_persistent
    DevBuf<uint32_t> refImage;                 // fyprt_compare_image's reference
    DevBuf<float4> shadowTasks; DevBuf<uint32_t> queueCounters, sortCounts, sortOffset, sortTotal, sortIndex; DevBuf<uint8_t> sortKeys; DevBuf<uint16_t> sortHist;

    int fail(int code, const std::string& m) { err = m; return code; }
    int hip(hipError_t e, const char* what) {
        if (e == hipSuccess) return FYPRT_OK;
        err = std::string(what) + ": " + hipGetErrorString(e);
        std::fprintf(stderr, "fyprt: %s\n", err.c_str());     // the reference prints and continues (Renderer.cu:29-47)
        return FYPRT_EHIP;
    }
};

#define HIPCHK(ctx, call) do { int _rc = (ctx)->hip((call), #call); if (_rc != FYPRT_OK) return _rc; } while (0)

static hipError_t sync_all(fyprt_context* c) {      // both streams: the front one only ever runs ahead of `stream`
    hipError_t e = c->front ? hipStreamSynchronize(c->front) : hipSuccess;
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    return e != hipSuccess ? e : e2;
}

// Effective pending-entry budget of node_step's stack rule for the uploaded tree: tuning key 8 if set; otherwise a few entries
// above the tree's level count, rounded DOWN to a stack size at which one more 256-thread workgroup fits the CU's 160 KB of
// LDS (entries x 1 KB per workgroup: 32 -> 5 workgroups, 26 -> 6, 22 -> 7, 20 -> 8, 17 -> 9, 16 -> 10) as long as at least 4
// entries of slack above the level count remain.  Never below the level count (the induction needs it), never above 31.
static int effective_stack_budget(const fyprt_context* c) {
    const int levels = (int)c->hostBvh.levels;
    if (c->tuning[8] > 0) return std::min((int)rth::kStackBudget, std::max(levels, c->tuning[8]));
    static const int kSizes[6] = {16, 17, 20, 22, 26, 32};
    int entries = 32;
    for (int i = 5; i >= 0; --i) if (kSizes[i] <= std::max(levels + 9, 16) && kSizes[i] - 1 >= levels + 4) { entries = kSizes[i]; break; }
    return std::min((int)rth::kStackBudget, std::max(levels, entries - 1));
}

extern "C" {

void fyprt_comm_destroy(fyprt_context* c);

const char* fyprt_version(void) { return "fyprt 0.1.0 gfx950 (wave64, LDS traversal stack, fp-contract off)"; }

int fyprt_create(int device_ordinal, fyprt_context** out) {
    if (!out) { g_createError = "fyprt_create: out is NULL"; return FYPRT_EINVAL; }
    *out = nullptr;
    if (device_ordinal == -1) {          // host-only context: scene ingestion + builders + exports, no device (CPU tests)
        auto* hc = new fyprt_context(); hc->device = -1; hc->hostOnly = true; *out = hc; return FYPRT_OK;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) { g_createError = std::string("no HIP device: ") + hipGetErrorString(e); return FYPRT_EHIP; }
    if (device_ordinal < 0 || device_ordinal >= count) { g_createError = "device ordinal out of range"; return FYPRT_EINVAL; }
    e = hipSetDevice(device_ordinal);
    if (e != hipSuccess) { g_createError = std::string("hipSetDevice: ") + hipGetErrorString(e); return FYPRT_EHIP; }
    auto* c = new fyprt_context(); c->device = device_ordinal;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_createError = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return FYPRT_EHIP; }
    {   // The front stream is created at the LOWEST priority: HIP deals the streams of one priority level round-robin over a
        // few hardware queues (GPU_MAX_HW_QUEUES, default 4), and two streams that land on the same queue run strictly one
        // after the other.  With torch + RCCL streams in the process both of ours shared a queue and nothing overlapped
        // (1.12 instead of 0.99 ms per frame); a different priority level uses a different set of queues (1.00 ms), and with
        // GPU_MAX_HW_QUEUES=8 in the environment as well, 0.98 ms with or without the per-frame gather (profiles/README.md).
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&c->front, hipStreamNonBlocking, lo);
    }
    for (int k = 0; k < 2 && e == hipSuccess; ++k) { e = hipEventCreateWithFlags(&c->evFront[k], hipEventDisableTiming); if (e == hipSuccess) e = hipEventCreateWithFlags(&c->evDone[k], hipEventDisableTiming); }
    if (e != hipSuccess) { g_createError = std::string("front stream / events: ") + hipGetErrorString(e); delete c; return FYPRT_EHIP; }
    for (auto& row : c->ring) for (auto& e : row) (void)hipEventCreate(&e);
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) c->numCUs = prop.multiProcessorCount; }
    if (const char* e = std::getenv("FYPRT_TOP_NODES")) c->tuning[16] = std::min(1024, std::max(0, std::atoi(e)));
    (void)c->queueCounters.alloc(8);          // two queues (frame parity): tail, head, pad, pad each
    (void)c->rayCounter.alloc(32);            // 4 launches x (rays, box tests, triangle tests, hits, node visits, 3 unused)
    (void)hipMemset(c->rayCounter.p, 0, 256);
    *out = c;
    return FYPRT_OK;
}

void fyprt_destroy(fyprt_context* c) {
    if (!c) return;
    if (c->hostOnly) { delete c; return; }
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    fyprt_comm_destroy(c);
    c->accum.release(); c->image.release(); c->payload.release(); c->depth.release(); c->normalA.release(); c->normalB.release();
    c->di.release(); c->diPrev.release(); c->gi.release(); c->giPrev.release(); c->giHot.release(); c->drec.release(); c->dprevA.release(); c->dprevB.release();
    c->nodes.release(); c->leafTris.release(); c->triPos.release(); c->triShade.release(); c->mats.release(); c->texTable.release();
    for (auto& t : c->texPixels) t.release();
    c->emissive.release(); c->lightRecs.release(); c->ltTlas.release(); c->ltBlas.release(); c->ltFirst.release(); c->ltCount.release(); c->ltRoot.release(); c->ltLeafOfTri.release();
    c->rayCounter.release(); c->shadowTasks.release(); c->queueCounters.release(); c->refImage.release(); c->objVerts.release();
    for (int k = 0; k < 2; ++k) { c->wfRays[k].release(); c->wfHits[k].release(); }
    c->wfState.release(); c->wfPixels.release(); c->wfPixels2.release(); c->wfCounters.release();
    c->sortCounts.release(); c->sortOffset.release(); c->sortTotal.release(); c->sortIndex.release(); c->sortKeys.release(); c->sortHist.release();
    for (auto& row : c->ring) for (auto& e : row) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) { if (c->evFront[k]) (void)hipEventDestroy(c->evFront[k]); if (c->evDone[k]) (void)hipEventDestroy(c->evDone[k]); }
    if (c->front) (void)hipStreamDestroy(c->front);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* fyprt_last_error(const fyprt_context* c) { return c ? c->err.c_str() : g_createError.c_str(); }

int fyprt_resize(fyprt_context* c, uint32_t w, uint32_t h) {
    if (!c) return FYPRT_EINVAL;
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 31)) return c->fail(FYPRT_EINVAL, "fyprt_resize: bad size");
    if (c->hostOnly) return c->fail(FYPRT_ESTATE, "host-only context (device -1) has no device buffers");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    const size_t n = (size_t)w * h;
    HIPCHK(c, c->accum.alloc(n)); HIPCHK(c, c->image.alloc(n)); HIPCHK(c, c->payload.alloc(n)); HIPCHK(c, c->depth.alloc(n));
    HIPCHK(c, c->normalA.alloc(n)); HIPCHK(c, c->normalB.alloc(n));
    HIPCHK(c, c->di.alloc(n)); HIPCHK(c, c->diPrev.alloc(n)); HIPCHK(c, c->gi.alloc(n)); HIPCHK(c, c->giPrev.alloc(n)); HIPCHK(c, c->giHot.alloc(n * 4));
    HIPCHK(c, hipMemsetAsync(c->giHot.p, 0, c->giHot.bytes(), c->stream));
    HIPCHK(c, c->drec.alloc(n)); HIPCHK(c, c->dprevA.alloc(n)); HIPCHK(c, c->dprevB.alloc(n));
    HIPCHK(c, hipMemsetAsync(c->drec.p, 0, c->drec.bytes(), c->stream)); HIPCHK(c, hipMemsetAsync(c->dprevA.p, 0, c->dprevA.bytes(), c->stream));
    HIPCHK(c, hipMemsetAsync(c->dprevB.p, 0, c->dprevB.bytes(), c->stream));
    {   // shadow-task storage: 256 slots per setup workgroup (grid padded to whole groups of 8 tile rows), + the sort scratch
        const size_t tilesX = (w + 15u) / 16u, tilesY = (h + 15u) / 16u;
        const size_t maxGroups = std::max(tilesX * ((tilesY + 7u) / 8u) * 8u, ((tilesX * tilesY + 7u) / 8u) * 8u);
        HIPCHK(c, c->shadowTasks.alloc((size_t)maxGroups * 256u * 4u * 2u));   // two queues (frame parity) of 64-byte tasks
        c->queueStride = (size_t)maxGroups * 256u * 4u;
        // the sort scratch alternates with the frame parity like the queues: frame N+1's setup / scan / scatter (front stream) run
        // beside frame N's trace kernel, which still reads its `sorted` index
        c->sortGroups = maxGroups;
        HIPCHK(c, c->sortCounts.alloc(2 * maxGroups)); HIPCHK(c, c->sortKeys.alloc(2 * maxGroups * 256u)); HIPCHK(c, c->sortHist.alloc(2 * maxGroups * kSortBins));
        HIPCHK(c, c->sortOffset.alloc(2 * maxGroups * kSortBins)); HIPCHK(c, c->sortTotal.alloc(2 * kSortBins)); HIPCHK(c, c->sortIndex.alloc(2 * maxGroups * 256u));
    }
    // cudaMemset(…, 0, …) of every buffer: Renderer.cu:333-355, :372, :393, :414
    HIPCHK(c, hipMemsetAsync(c->accum.p, 0, c->accum.bytes(), c->stream)); HIPCHK(c, hipMemsetAsync(c->image.p, 0, c->image.bytes(), c->stream));
    HIPCHK(c, hipMemsetAsync(c->payload.p, 0, c->payload.bytes(), c->stream)); HIPCHK(c, hipMemsetAsync(c->depth.p, 0, c->depth.bytes(), c->stream));
    HIPCHK(c, hipMemsetAsync(c->normalA.p, 0, c->normalA.bytes(), c->stream)); HIPCHK(c, hipMemsetAsync(c->normalB.p, 0, c->normalB.bytes(), c->stream));
    HIPCHK(c, hipMemsetAsync(c->di.p, 0, c->di.bytes(), c->stream)); HIPCHK(c, hipMemsetAsync(c->diPrev.p, 0, c->diPrev.bytes(), c->stream));
    HIPCHK(c, hipMemsetAsync(c->gi.p, 0, c->gi.bytes(), c->stream)); HIPCHK(c, hipMemsetAsync(c->giPrev.p, 0, c->giPrev.bytes(), c->stream));
    HIPCHK(c, sync_all(c));
    c->part1Pending = false;
    c->stripeRows = 0; c->stripeParts = 1; c->stripePart = 0;
    c->W = w; c->H = h; c->frameIndex = 1; c->normalFlip = false; c->dprevFlip = false; c->lastTech = -1; c->lastRestir = -1; c->externalImage = nullptr;
    c->histDI[0] = c->histGI[0] = 0; c->histDI[1] = c->histGI[1] = h;        // zero-filled history: "valid" everywhere, M = 0
    if (!c->rowsSet || c->rowEnd > h) { c->rowBegin = 0; c->rowEnd = h; c->halo = 0; c->rowsSet = false; }
    return FYPRT_OK;
}

// rows of the frame a striped context owns: the stripes k * parts + part, the last one cut at the frame's end
static uint32_t stripe_row_count(uint32_t H, uint32_t stripe, uint32_t parts, uint32_t part) {
    uint32_t n = 0;
    for (uint64_t r0 = (uint64_t)part * stripe; r0 < H; r0 += (uint64_t)parts * stripe) n += (uint32_t)std::min<uint64_t>(stripe, H - r0);
    return n;
}

int fyprt_set_rows(fyprt_context* c, uint32_t b, uint32_t e, uint32_t halo) {
    if (!c) return FYPRT_EINVAL;
    if (c->H == 0) return c->fail(FYPRT_ESTATE, "fyprt_set_rows before fyprt_resize");
    if (b >= e || e > c->H) return c->fail(FYPRT_EINVAL, "fyprt_set_rows: need row_begin < row_end <= height");
    c->rowBegin = b; c->rowEnd = e; c->halo = halo; c->rowsSet = true; c->stripeRows = 0;
    return FYPRT_OK;
}
// The interleaved split of SURVEY.md §8(e) for the techniques whose pixels are independent (0-6): the frame is cut into stripes of
// `stripe_rows` rows and this context renders every `parts`-th of them, starting with stripe `part` — the stripes of one context
// sample the whole image, so the parts cost the same without any balancing.  stripe_rows 0 returns to the band of fyprt_set_rows
// (whole frame if none was set).  ReSTIR frames refuse a striped context: spatial reuse reads the rows around a pixel.
int fyprt_set_row_stripes(fyprt_context* c, uint32_t stripe_rows, uint32_t parts, uint32_t part) {
    if (!c) return FYPRT_EINVAL;
    if (c->H == 0) return c->fail(FYPRT_ESTATE, "fyprt_set_row_stripes before fyprt_resize");
    if (stripe_rows == 0) { c->stripeRows = 0; c->stripeParts = 1; c->stripePart = 0; return FYPRT_OK; }
    if (parts == 0 || part >= parts) return c->fail(FYPRT_EINVAL, "fyprt_set_row_stripes: need part < parts");
    if (stripe_row_count(c->H, stripe_rows, parts, part) == 0) return c->fail(FYPRT_EINVAL, "fyprt_set_row_stripes: this part owns no row (fewer stripes than parts)");
    c->stripeRows = stripe_rows; c->stripeParts = parts; c->stripePart = part;
    c->rowBegin = 0; c->rowEnd = c->H; c->halo = 0; c->rowsSet = false;
    return FYPRT_OK;
}

static int upload(fyprt_context* c, void* dst, const void* src, size_t bytes) {
    if (bytes == 0 || c->hostOnly) return FYPRT_OK;
    return c->hip(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
}


// the node array back from the device, into HOST form (rt_host.h: inner references are indices on the host, byte offsets on the device)
static int download_nodes(fyprt_context* c) {
    if (c->hostBvh.nodes.empty()) return FYPRT_OK;
    HIPCHK(c, hipMemcpy(c->hostBvh.nodes.data(), c->nodes.p, c->hostBvh.nodes.size() * 64, hipMemcpyDeviceToHost));
    rth::nodes_to_host_form(c->hostBvh.nodes.data(), c->hostBvh.nodes.size());
    return FYPRT_OK;
}

// the refit pass over the whole tree (rt_refit.h): leaf triangles from the per-triangle positions, then boxes + quantisation
// level by level, bottom level first
static int run_refit(fyprt_context* c) {
    const uint32_t nLeaf = (uint32_t)c->hostBvh.tris.size(), nNodes = (uint32_t)c->hostBvh.nodes.size();
    if (nLeaf) hipLaunchKernelGGL(k_refresh_leaf_tris, dim3((nLeaf + 255u) / 256u), dim3(256), 0, c->stream, c->triPos.p, c->leafTris.p, nLeaf);
    for (uint32_t l = 1; nNodes && l <= c->hostBvh.levels; ++l) {
        const uint32_t first = c->levelOffset[l], count = c->levelOffset[l + 1] - first;
        if (count) hipLaunchKernelGGL(k_refit_level, dim3((count + 127u) / 128u), dim3(128), 0, c->stream, c->nodes.p, c->levelNodes.p + first, count, c->leafTris.p, c->triPos.p, c->nodeBox.p);
    }
    HIPCHK(c, hipGetLastError());
    return FYPRT_OK;
}

// Device builder (tuning key 12, rt_lbvh.h): Morton keys -> radix sort -> Karras radix tree -> BFS collapse into 4-wide nodes ->
// level counts.  Leaves c->nodes (topology only), c->leafTris (triangle indices in sorted order) and the host copy of the
// topology (for export and the level grouping); boxes come from run_refit.  kLbvhTooDeep: more than 31 wide levels.
constexpr int kLbvhTooDeep = -1000;
static int env_int(const char* name, int fallback) { const char* v = std::getenv(name); return (v && *v) ? std::atoi(v) : fallback; }
static int build_device_lbvh(fyprt_context* c, const fyprt_vertex* verts, uint32_t nV, uint32_t nT, bool ploc) {
    float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (uint32_t i = 0; i < nV; ++i) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], verts[i].position[a]); hi[a] = std::max(hi[a], verts[i].position[a]); }
    float3 l3 = make_float3(lo[0], lo[1], lo[2]), ie = make_float3(hi[0] > lo[0] ? 1.0f / (hi[0] - lo[0]) : 0.0f, hi[1] > lo[1] ? 1.0f / (hi[1] - lo[1]) : 0.0f, hi[2] > lo[2] ? 1.0f / (hi[2] - lo[2]) : 0.0f);
    struct Temps {                      // scratch of the build, freed on every way out
        DevBuf<unsigned long long> keysA, keysB; DevBuf<uint32_t> valsA, valsB, parentOfNode, parentOfLeaf, arrived; DevBuf<float> box; DevBuf<RadixNode> radix; DevBuf<CollapseItem> qA, qB; DevBuf<uint32_t> counters; DevBuf<float4> wide; DevBuf<uint8_t> temp;
        DevBuf<uint32_t> clA, clB, nearest; DevBuf<uint8_t> keep, selTemp;                          // PLOC
        ~Temps() { clA.release(); clB.release(); nearest.release(); keep.release(); selTemp.release(); keysA.release(); keysB.release(); valsA.release(); valsB.release(); parentOfNode.release(); parentOfLeaf.release(); arrived.release(); box.release(); radix.release(); qA.release(); qB.release(); counters.release(); wide.release(); temp.release(); }
    } t;
    auto &keysA = t.keysA, &keysB = t.keysB; auto &valsA = t.valsA, &valsB = t.valsB; auto& radix = t.radix;
    if (!ploc) {
        HIPCHK(c, t.parentOfNode.alloc(nT)); HIPCHK(c, t.parentOfLeaf.alloc(nT)); HIPCHK(c, t.arrived.alloc(nT)); HIPCHK(c, t.box.alloc((size_t)nT * 6));
        HIPCHK(c, hipMemsetAsync(t.arrived.p, 0, (size_t)nT * 4, c->stream));
    } else {
        HIPCHK(c, t.box.alloc((size_t)nT * 12)); HIPCHK(c, t.clA.alloc(nT)); HIPCHK(c, t.clB.alloc(nT)); HIPCHK(c, t.nearest.alloc(nT)); HIPCHK(c, t.keep.alloc(nT));
    } auto &qA = t.qA, &qB = t.qB; auto& counters = t.counters; auto& wide = t.wide; auto& temp = t.temp;
    HIPCHK(c, keysA.alloc(nT)); HIPCHK(c, keysB.alloc(nT)); HIPCHK(c, valsA.alloc(nT)); HIPCHK(c, valsB.alloc(nT)); HIPCHK(c, radix.alloc(nT)); HIPCHK(c, qA.alloc(nT)); HIPCHK(c, qB.alloc(nT));
    HIPCHK(c, counters.alloc(4)); HIPCHK(c, wide.alloc((size_t)nT * 4));
    hipLaunchKernelGGL(k_lbvh_keys, dim3((nT + 255u) / 256u), dim3(256), 0, c->stream, c->triPos.p, nT, l3, ie, keysA.p, valsA.p);
    size_t tempBytes = 0;
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, keysA.p, keysB.p, valsA.p, valsB.p, (int)nT, 0, 63, c->stream));
    HIPCHK(c, temp.alloc(tempBytes));
    HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(temp.p, tempBytes, keysA.p, keysB.p, valsA.p, valsB.p, (int)nT, 0, 63, c->stream));
    uint32_t rootNode = 0;
    if (!ploc) {
        hipLaunchKernelGGL(k_lbvh_radix, dim3((nT + 255u) / 256u), dim3(256), 0, c->stream, keysB.p, (int)nT, radix.p, t.parentOfNode.p, t.parentOfLeaf.p);
        hipLaunchKernelGGL(k_lbvh_boxes, dim3((nT + 255u) / 256u), dim3(256), 0, c->stream, radix.p, t.parentOfNode.p, t.parentOfLeaf.p, valsB.p, c->triPos.p, nT, t.arrived.p, t.box.p);
    } else {
        // PLOC rounds over the sorted order (rt_lbvh.h).  Every round has at least one mutual pair (the pair of globally smallest
        // union), real scenes lose 20-35 % of their clusters per round (54 rounds for 1 M triangles).  A round over a LONG list
        // that loses less than 1/16 is followed by a forced one (neighbours i, i ^ 1 merge), which halves the list, so degenerate
        // input cannot take O(n) rounds of O(n) work; short lists (the top of the tree, where quality counts most) are never forced.
        const int radius = std::min(kPlocMaxRadius, std::max(1, env_int("FYPRT_PLOC_RADIUS", 16)));
        hipLaunchKernelGGL(k_ploc_leaves, dim3((nT + 255u) / 256u), dim3(256), 0, c->stream, valsB.p, c->triPos.p, nT, t.box.p, t.clA.p);
        HIPCHK(c, hipMemsetAsync(counters.p + 2, 0, 8, c->stream));                     // [2] binary nodes allocated, [3] clusters kept by the compaction
        size_t selBytes = 0;
        HIPCHK(c, hipcub::DeviceSelect::Flagged(nullptr, selBytes, t.clB.p, t.keep.p, t.clA.p, counters.p + 3, (int)nT, c->stream));
        auto& selTemp = t.selTemp; HIPCHK(c, selTemp.alloc(selBytes));
        uint32_t n = nT; uint32_t* cur = t.clA.p; uint32_t* merged = t.clB.p; int force = 0, rounds = 0;
        while (n > 1) {
            const dim3 g((n + kPlocBlock - 1) / kPlocBlock);
            hipLaunchKernelGGL(k_ploc_nearest, g, dim3(kPlocBlock), 0, c->stream, (const uint32_t*)cur, n, (const float*)t.box.p, nT, radius, force, t.nearest.p);
            hipLaunchKernelGGL(k_ploc_merge, g, dim3(kPlocBlock), 0, c->stream, (const uint32_t*)cur, (const uint32_t*)t.nearest.p, n, nT, radix.p, t.box.p, counters.p + 2, merged, t.keep.p);
            hipError_t e = hipcub::DeviceSelect::Flagged(selTemp.p, selBytes, merged, t.keep.p, cur, counters.p + 3, (int)n, c->stream);     // back into `cur`, order kept
            uint32_t kept = 0;
            if (e == hipSuccess) e = hipMemcpyAsync(&kept, counters.p + 3, 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess || kept == 0 || kept >= n) return e != hipSuccess ? c->hip(e, "PLOC round") : c->fail(FYPRT_EHIP, "PLOC round made no progress");
            if (env_int("FYPRT_BVH_DEBUG", 0)) std::fprintf(stderr, "PLOC round %d: %u -> %u clusters%s\n", rounds, n, kept, force ? " (forced)" : "");
            force = (!force && n > 4096u && kept > n - n / 16u) ? 1 : 0;
            n = kept; ++rounds;
        }
        HIPCHK(c, hipMemcpy(&rootNode, cur, 4, hipMemcpyDeviceToHost));
        c->lastBuildRounds = rounds;
    }
    HIPCHK(c, c->leafTris.alloc((size_t)nT * 3));
    // BFS collapse, one launch per level; the nodes of a level are contiguous: [levelFirst[l], levelFirst[l + 1])
    const CollapseItem rootItem{rootNode, 0u, 0u};
    uint32_t h_counters[2] = {1u, 0u};
    HIPCHK(c, hipMemcpyAsync(qA.p, &rootItem, sizeof rootItem, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(counters.p, h_counters, 8, hipMemcpyHostToDevice, c->stream));
    std::vector<uint32_t> levelFirst{0u};
    uint32_t nIn = 1, total = 1;
    CollapseItem *in = qA.p, *out = qB.p;
    while (nIn) {
        hipLaunchKernelGGL(k_lbvh_collapse, dim3((nIn + 127u) / 128u), dim3(128), 0, c->stream, (const RadixNode*)radix.p, (const float*)t.box.p, (const uint32_t*)valsB.p, (const CollapseItem*)in, nIn, out, counters.p, wide.p, c->leafTris.p);
        HIPCHK(c, hipMemcpyAsync(h_counters, counters.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        levelFirst.push_back(total);
        nIn = h_counters[1]; total = h_counters[0];
        h_counters[1] = 0;
        HIPCHK(c, hipMemcpyAsync(counters.p + 1, &h_counters[1], 4, hipMemcpyHostToDevice, c->stream));
        std::swap(in, out);
        if (levelFirst.size() > rth::kStackBudget + 1u) return kLbvhTooDeep;
    }
    const uint32_t nLevels = (uint32_t)levelFirst.size() - 1u;                      // BFS levels = wide levels of the tree
    for (uint32_t l = nLevels; l-- > 0;) {
        const uint32_t first = levelFirst[l], count = levelFirst[l + 1] - first;
        if (count) hipLaunchKernelGGL(k_lbvh_levels, dim3((count + 127u) / 128u), dim3(128), 0, c->stream, wide.p, first, count);
    }
    HIPCHK(c, hipGetLastError());
    if (total >= rth::kMaxNodes) return c->fail(FYPRT_EINVAL, "fyprt_upload_scene: too many nodes for the node reference range");
    HIPCHK(c, c->nodes.alloc((size_t)total * 4));
    HIPCHK(c, hipMemcpyAsync(c->nodes.p, wide.p, (size_t)total * 64, hipMemcpyDeviceToDevice, c->stream));
    rth::SceneBVH& b = c->hostBvh; b = rth::SceneBVH();
    b.nodes.resize(total); b.tris.resize(nT); b.rootRef = 0; b.levels = nLevels;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return download_nodes(c);                                                                        // topology + meta: the level grouping needs it
}

int fyprt_upload_scene(fyprt_context* c, const fyprt_scene_desc* s) {
    if (!c || !s) return FYPRT_EINVAL;
    if ((s->triangle_count && (!s->triangles || !s->vertices || s->triangle_stride < 16)) || (s->mesh_count && !s->meshes) ||
        (s->material_count && !s->materials))
        return c->fail(FYPRT_EINVAL, "fyprt_upload_scene: NULL array with non-zero count");
    if (!c->hostOnly) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c)); }
    const uint8_t* tb = (const uint8_t*)s->triangles;
    auto tri = [&](uint32_t i) { return reinterpret_cast<const uint32_t*>(tb + (size_t)i * s->triangle_stride); };
    for (uint32_t i = 0; i < s->triangle_count; ++i) {
        const uint32_t* t = tri(i);
        if (t[0] >= s->vertex_count || t[1] >= s->vertex_count || t[2] >= s->vertex_count || (int32_t)t[3] < 0 || t[3] >= s->material_count)
            return c->fail(FYPRT_EINVAL, "fyprt_upload_scene: triangle " + std::to_string(i) + " references a vertex/material out of range");
    }
    uint64_t covered = 0;
    for (uint32_t m = 0; m < s->mesh_count; ++m) {
        const fyprt_mesh& me = s->meshes[m];
        if ((uint64_t)me.first_triangle + me.triangle_count > s->triangle_count || me.material_index < 0 || (uint32_t)me.material_index >= s->material_count)
            return c->fail(FYPRT_EINVAL, "fyprt_upload_scene: mesh " + std::to_string(m) + " range/material out of bounds");
        covered += me.triangle_count;
    }
    if (covered != s->triangle_count) return c->fail(FYPRT_EINVAL, "fyprt_upload_scene: meshes must partition the triangle list");
    struct HostOnlyGuard { bool prev; explicit HostOnlyGuard(bool on) : prev(g_hostOnlyAlloc) { g_hostOnlyAlloc = on; } ~HostOnlyGuard() { g_hostOnlyAlloc = prev; } } guard(c->hostOnly);
    // acceleration structure (ours): built on the host (binned SAH + collapse, bvh_build.cpp) or, with tuning key 12, on the
    // device (LBVH + collapse, rt_lbvh.h — further down, once the per-triangle records are on the device)
    bool deviceBuild = !c->hostOnly && c->tuning[12] != 0 && s->triangle_count > 4;
    auto hostBuild = [&]() -> int {
        rth::BuildSceneBVH(s->vertices, tb, s->triangle_stride, s->meshes, s->mesh_count, c->hostBvh);
        if (c->hostBvh.levels > rth::kStackBudget || c->hostBvh.nodes.size() >= (size_t)rth::kMaxNodes)
            return c->fail(FYPRT_EINVAL, "fyprt_upload_scene: acceleration structure (" + std::to_string(c->hostBvh.levels) + " levels, " +
                           std::to_string(c->hostBvh.nodes.size()) + " nodes) exceeds the traversal stack / node index range");
        HIPCHK(c, c->nodes.alloc(c->hostBvh.nodes.size() * 4)); HIPCHK(c, c->leafTris.alloc(c->hostBvh.tris.size() * 3));
        std::vector<rth::Node> dn;                                        // the array in device form (inner references = byte offsets)
        if (!c->hostOnly) { dn = c->hostBvh.nodes; rth::nodes_to_device_form(dn.data(), dn.size()); }
        if (upload(c, c->nodes.p, dn.data(), c->nodes.bytes()) || upload(c, c->leafTris.p, c->hostBvh.tris.data(), c->leafTris.bytes())) return FYPRT_EHIP;
        return FYPRT_OK;
    };
    if (!deviceBuild) { const int rc = hostBuild(); if (rc != FYPRT_OK) return rc; }
    // per-triangle gather records
    const uint32_t nT = s->triangle_count;
    std::vector<float> pos((size_t)nT * 12), shade((size_t)nT * 16);
    for (uint32_t i = 0; i < nT; ++i) {
        const uint32_t* t = tri(i);
        const fyprt_vertex &a = s->vertices[t[0]], &b = s->vertices[t[1]], &cc = s->vertices[t[2]];
        float* p = &pos[(size_t)i * 12]; float* q = &shade[(size_t)i * 16];
        float matBits; std::memcpy(&matBits, &t[3], 4);
        p[0] = a.position[0]; p[1] = a.position[1]; p[2] = a.position[2]; p[3] = matBits;
        p[4] = b.position[0]; p[5] = b.position[1]; p[6] = b.position[2]; p[7] = 0.0f;
        p[8] = cc.position[0]; p[9] = cc.position[1]; p[10] = cc.position[2]; p[11] = 0.0f;
        q[0] = a.normal[0]; q[1] = a.normal[1]; q[2] = a.normal[2]; q[3] = a.uv[0];
        q[4] = b.normal[0]; q[5] = b.normal[1]; q[6] = b.normal[2]; q[7] = a.uv[1];
        q[8] = cc.normal[0]; q[9] = cc.normal[1]; q[10] = cc.normal[2]; q[11] = b.uv[0];
        q[12] = b.uv[1]; q[13] = cc.uv[0]; q[14] = cc.uv[1]; q[15] = matBits;
    }
    HIPCHK(c, c->triPos.alloc((size_t)nT * 3)); HIPCHK(c, c->triShade.alloc((size_t)nT * 4));
    if (upload(c, c->triPos.p, pos.data(), c->triPos.bytes()) || upload(c, c->triShade.p, shade.data(), c->triShade.bytes())) return FYPRT_EHIP;
    // refit support: vertices + per-triangle indices on the device, nodes grouped by level (levels = 1 first)
    c->vertexCount = s->vertex_count; c->hostBvhStale = false;
    c->hostVerts.assign(s->vertices, s->vertices + s->vertex_count); c->objVerts.release(); c->meshFirstVertex.clear();
    c->topoTris.resize((size_t)nT * 4);
    for (uint32_t i = 0; i < nT; ++i) std::memcpy(&c->topoTris[(size_t)i * 4], tri(i), 16);
    if (deviceBuild) {
        const int rc = build_device_lbvh(c, s->vertices, s->vertex_count, nT, c->tuning[12] == 2);
        if (rc == kLbvhTooDeep) { deviceBuild = false; const int rc2 = hostBuild(); if (rc2 != FYPRT_OK) return rc2; }   // > 31 wide levels: the host builder bounds them
        else if (rc != FYPRT_OK) return rc;
    }
    c->topoMeshes.assign(s->meshes, s->meshes + s->mesh_count); c->topoMats.assign(s->materials, s->materials + s->material_count);
    c->prebuiltLightTrees = s->light_trees && s->light_trees->tlas_nodes;
    {
        const std::vector<rth::Node>& hn = c->hostBvh.nodes;
        const uint32_t L = c->hostBvh.levels;
        c->levelOffset.assign(L + 2, 0);
        for (const rth::Node& n : hn) c->levelOffset[(n.meta >> 3) + 1]++;
        for (uint32_t l = 1; l <= L + 1; ++l) c->levelOffset[l] += c->levelOffset[l - 1];      // levelOffset[l] = first slot of level l (1-based levels)
        std::vector<uint32_t> order(hn.size()), fill(c->levelOffset.begin(), c->levelOffset.end());
        for (uint32_t i = 0; i < (uint32_t)hn.size(); ++i) order[fill[hn[i].meta >> 3]++] = i;
        HIPCHK(c, c->dverts.alloc(s->vertex_count)); HIPCHK(c, c->triIdx.alloc(nT)); HIPCHK(c, c->levelNodes.alloc(order.size())); HIPCHK(c, c->nodeBox.alloc(hn.size() * 2));
        if (upload(c, c->dverts.p, s->vertices, c->dverts.bytes()) || upload(c, c->triIdx.p, c->topoTris.data(), c->triIdx.bytes()) ||
            upload(c, c->levelNodes.p, order.data(), c->levelNodes.bytes())) return FYPRT_EHIP;
    }
    if (deviceBuild) {                     // the device builder leaves boxes and quantisation to the refit pass
        const int rc = run_refit(c);
        if (rc != FYPRT_OK) return rc;
        HIPCHK(c, sync_all(c));
        { const int rc2 = download_nodes(c); if (rc2 != FYPRT_OK) return rc2; }
        HIPCHK(c, hipMemcpy(c->hostBvh.tris.data(), c->leafTris.p, c->hostBvh.tris.size() * 48, hipMemcpyDeviceToHost));
    }
    // materials (Material.cuh:7-16 -> 3 quads)
    std::vector<float> mats((size_t)s->material_count * 12, 0.0f);
    std::vector<char> emissiveMat(s->material_count, 0);
    for (uint32_t i = 0; i < s->material_count; ++i) {
        const fyprt_material& m = s->materials[i]; float* q = &mats[(size_t)i * 12];
        q[0] = m.albedo[0]; q[1] = m.albedo[1]; q[2] = m.albedo[2];
        uint32_t info = ((m.is_use_albedo_map & 0xFFu) ? 0x80000000u : 0u) | (m.albedo_map_index > 0x7FFFFFFFu ? 0x7FFFFFFFu : m.albedo_map_index);
        std::memcpy(&q[3], &info, 4);
        q[4] = m.roughness; q[5] = m.metallic; q[6] = m.emission_power;
        q[8] = m.emission_color[0]; q[9] = m.emission_color[1]; q[10] = m.emission_color[2];
        const float ex = m.emission_color[0] * m.emission_power, ey = m.emission_color[1] * m.emission_power, ez = m.emission_color[2] * m.emission_power;
        emissiveMat[i] = ((ex * ex + ey * ey) + ez * ez) > 0.0f;       // glm::length2(GetEmission()) > 0 (Scene.cpp:216)
    }
    HIPCHK(c, c->mats.alloc((size_t)s->material_count * 3));
    if (upload(c, c->mats.p, mats.data(), c->mats.bytes())) return FYPRT_EHIP;
    // textures
    for (auto& t : c->texPixels) t.release();
    c->texPixels.assign(s->texture_count, DevBuf<uint32_t>());
    std::vector<DevTexture> tt(s->texture_count);
    for (uint32_t i = 0; i < s->texture_count; ++i) {
        const fyprt_texture& t = s->textures[i];
        if (!t.pixels || t.width == 0 || t.height == 0) return c->fail(FYPRT_EINVAL, "fyprt_upload_scene: empty texture");
        HIPCHK(c, c->texPixels[i].alloc((size_t)t.width * t.height));
        if (upload(c, c->texPixels[i].p, t.pixels, c->texPixels[i].bytes())) return FYPRT_EHIP;
        tt[i] = DevTexture{c->texPixels[i].p, t.width, t.height, 0};
    }
    HIPCHK(c, c->texTable.alloc(s->texture_count));
    if (upload(c, c->texTable.p, tt.data(), c->texTable.bytes())) return FYPRT_EHIP;
    // emissive list (Scene::InitSceneEmissiveTriangles, Scene.cpp:209-221)
    std::vector<uint32_t> em;
    if (s->emissive_triangles) em.assign(s->emissive_triangles, s->emissive_triangles + s->emissive_count);
    else for (uint32_t i = 0; i < nT; ++i) if (emissiveMat[tri(i)[3]]) em.push_back(i);
    for (uint32_t e : em) if (e >= nT) return c->fail(FYPRT_EINVAL, "fyprt_upload_scene: emissive triangle index out of range");
    HIPCHK(c, c->emissive.alloc(em.size()));
    if (upload(c, c->emissive.p, em.data(), c->emissive.bytes())) return FYPRT_EHIP;
    // light trees: prebuilt (reference shape) or ours
    rth::LightTrees& lt = c->hostLt; lt = rth::LightTrees();
    c->meshCount = s->mesh_count;
    if (s->light_trees && s->light_trees->tlas_nodes) {
        const fyprt_lighttrees& L = *s->light_trees;
        lt.tlas.assign(L.tlas_nodes, L.tlas_nodes + L.tlas_node_count); lt.tlasRoot = L.tlas_root;
        lt.first.assign(L.blas_first, L.blas_first + s->mesh_count); lt.count.assign(L.blas_count, L.blas_count + s->mesh_count);
        lt.root.assign(L.blas_root, L.blas_root + s->mesh_count);
        uint32_t total = 0; for (uint32_t m = 0; m < s->mesh_count; ++m) total = std::max(total, lt.first[m] + lt.count[m]);
        lt.blas.assign(L.blas_nodes, L.blas_nodes + total);
    } else {
        rth::BuildLightTrees(s->vertices, tb, s->triangle_stride, s->meshes, s->mesh_count, s->materials, lt);
    }
    static_assert(sizeof(DevLTNode) == sizeof(fyprt_lighttree_node), "light tree node layout");
    HIPCHK(c, c->ltTlas.alloc(lt.tlas.size())); HIPCHK(c, c->ltBlas.alloc(lt.blas.size()));
    HIPCHK(c, c->ltFirst.alloc(s->mesh_count)); HIPCHK(c, c->ltCount.alloc(s->mesh_count)); HIPCHK(c, c->ltRoot.alloc(s->mesh_count));
    if (upload(c, c->ltTlas.p, lt.tlas.data(), c->ltTlas.bytes()) || upload(c, c->ltBlas.p, lt.blas.data(), c->ltBlas.bytes()) ||
        upload(c, c->ltFirst.p, lt.first.data(), c->ltFirst.bytes()) || upload(c, c->ltCount.p, lt.count.data(), c->ltCount.bytes()) ||
        upload(c, c->ltRoot.p, lt.root.data(), c->ltRoot.bytes())) return FYPRT_EHIP;
    // ComputeDirectEmitterPMF (LightTree.cu:170-199) starts with a linear search for the first TLAS leaf whose mesh tree
    // holds the emitter; the answer does not depend on the shading point, so it is tabled per triangle here (same search
    // order: first match wins).
    std::vector<uint32_t> leafOfTri(nT, ~0u);
    for (uint32_t i = 0; i < (uint32_t)lt.tlas.size(); ++i) {
        if (!lt.tlas[i].is_leaf) continue;
        const uint32_t mesh = lt.tlas[i].right_or_emitter;
        if (mesh >= s->mesh_count) continue;
        for (uint32_t j = 0; j < lt.count[mesh]; ++j) {
            const fyprt_lighttree_node& n = lt.blas[lt.first[mesh] + j];
            if (n.is_leaf && n.right_or_emitter < nT && leafOfTri[n.right_or_emitter] == ~0u) leafOfTri[n.right_or_emitter] = i;
        }
    }
    HIPCHK(c, c->ltLeafOfTri.alloc(nT));
    if (upload(c, c->ltLeafOfTri.p, leafOfTri.data(), c->ltLeafOfTri.bytes())) return FYPRT_EHIP;
    DevScene& d = c->dsc;
    d.ltLeafOfTri = c->ltLeafOfTri.p;
    d.nodes = c->nodes.p; d.leafTris = c->leafTris.p; d.rootRef = rth::device_ref(c->hostBvh.rootRef); d.triCount = nT;
    d.triPos = c->triPos.p; d.triShade = c->triShade.p; d.mats = c->mats.p; d.textures = c->texTable.p; d.textureCount = s->texture_count;
    d.emissive = c->emissive.p; d.emissiveCount = (uint32_t)em.size();
    d.ltTlas = c->ltTlas.p; d.ltTlasCount = (uint32_t)lt.tlas.size(); d.ltTlasRoot = lt.tlasRoot;
    d.ltBlas = c->ltBlas.p; d.ltFirst = c->ltFirst.p; d.ltCount = c->ltCount.p; d.ltRoot = c->ltRoot.p;
    d.rayCounter = nullptr;
    // per-light records for ReSTIR DI, computed on the device with the kernels' own arithmetic
    HIPCHK(c, c->lightRecs.alloc(em.size() * 3));
    d.lightRecs = c->lightRecs.p;
    if (!c->hostOnly && !em.empty()) {
        hipLaunchKernelGGL(k_build_light_records, dim3(((uint32_t)em.size() + 255u) / 256u), dim3(256), 0, c->stream, d, c->lightRecs.p);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, sync_all(c));
    }
    c->haveScene = true;
    return FYPRT_OK;
}


// (Re)build the light trees from the given vertices with the stored topology and upload them (+ the emitter -> TLAS-leaf table).
static int rebuild_light_trees(fyprt_context* c, const fyprt_vertex* verts, const uint8_t* touched = nullptr) {
    rth::LightTrees& lt = c->hostLt;
    if (!touched) lt = rth::LightTrees();
    const uint32_t nT = (uint32_t)(c->topoTris.size() / 4), nM = (uint32_t)c->topoMeshes.size();
    rth::BuildLightTrees(verts, (const uint8_t*)c->topoTris.data(), 16, c->topoMeshes.data(), nM, c->topoMats.data(), lt, touched);
    HIPCHK(c, c->ltTlas.alloc(lt.tlas.size())); HIPCHK(c, c->ltBlas.alloc(lt.blas.size()));
    HIPCHK(c, c->ltFirst.alloc(nM)); HIPCHK(c, c->ltCount.alloc(nM)); HIPCHK(c, c->ltRoot.alloc(nM));
    if (upload(c, c->ltTlas.p, lt.tlas.data(), c->ltTlas.bytes()) || upload(c, c->ltBlas.p, lt.blas.data(), c->ltBlas.bytes()) ||
        upload(c, c->ltFirst.p, lt.first.data(), c->ltFirst.bytes()) || upload(c, c->ltCount.p, lt.count.data(), c->ltCount.bytes()) ||
        upload(c, c->ltRoot.p, lt.root.data(), c->ltRoot.bytes())) return FYPRT_EHIP;
    std::vector<uint32_t> leafOfTri(nT, ~0u);
    for (uint32_t i = 0; i < (uint32_t)lt.tlas.size(); ++i) {
        if (!lt.tlas[i].is_leaf) continue;
        const uint32_t mesh = lt.tlas[i].right_or_emitter;
        if (mesh >= nM) continue;
        for (uint32_t j = 0; j < lt.count[mesh]; ++j) {
            const fyprt_lighttree_node& n = lt.blas[lt.first[mesh] + j];
            if (n.is_leaf && n.right_or_emitter < nT && leafOfTri[n.right_or_emitter] == ~0u) leafOfTri[n.right_or_emitter] = i;
        }
    }
    HIPCHK(c, c->ltLeafOfTri.alloc(nT));
    if (upload(c, c->ltLeafOfTri.p, leafOfTri.data(), c->ltLeafOfTri.bytes())) return FYPRT_EHIP;
    DevScene& d = c->dsc;
    d.ltLeafOfTri = c->ltLeafOfTri.p; d.ltTlas = c->ltTlas.p; d.ltTlasCount = (uint32_t)lt.tlas.size(); d.ltTlasRoot = lt.tlasRoot;
    d.ltBlas = c->ltBlas.p; d.ltFirst = c->ltFirst.p; d.ltCount = c->ltCount.p; d.ltRoot = c->ltRoot.p;
    return FYPRT_OK;
}

// Scene geometry moved, topology unchanged (SceneManager::PerformAllSceneUpdates with a transform edit, SceneManager.cpp:24-66):
// new world vertices -> per-triangle records, leaf triangles and the tree's boxes are refreshed ON THE DEVICE (rt_refit.h),
// the per-light records are rebuilt by their kernel, the (small) light trees on the host.  The tree keeps its shape.
int fyprt_update_vertices(fyprt_context* c, const fyprt_vertex* vertices, uint32_t vertex_count) {
    if (!c || !vertices) return FYPRT_EINVAL;
    if (!c->haveScene) return c->fail(FYPRT_ESTATE, "fyprt_update_vertices before fyprt_upload_scene");
    if (vertex_count != c->vertexCount) return c->fail(FYPRT_EINVAL, "fyprt_update_vertices: vertex count differs from the uploaded scene");
    if (c->hostOnly) return c->fail(FYPRT_ESTATE, "fyprt_update_vertices needs a device (host-only context)");
    if (c->prebuiltLightTrees) return c->fail(FYPRT_ESTATE, "fyprt_update_vertices: the scene was uploaded with prebuilt light trees; upload it again instead");
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c));
    const uint32_t nT = (uint32_t)(c->topoTris.size() / 4);
    if (upload(c, c->dverts.p, vertices, c->dverts.bytes())) return FYPRT_EHIP;
    c->hostVerts.assign(vertices, vertices + vertex_count);
    if (nT) hipLaunchKernelGGL(k_refresh_triangles, dim3((nT + 255u) / 256u), dim3(256), 0, c->stream, c->dverts.p, c->triIdx.p, c->triPos.p, c->triShade.p, nT);
    { const int rr = run_refit(c); if (rr != FYPRT_OK) return rr; }
    c->hostBvhStale = true;
    int rc = rebuild_light_trees(c, vertices);
    if (rc != FYPRT_OK) return rc;
    if (c->dsc.emissiveCount) hipLaunchKernelGGL(k_build_light_records, dim3((c->dsc.emissiveCount + 255u) / 256u), dim3(256), 0, c->stream, c->dsc, c->lightRecs.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, sync_all(c));
    return FYPRT_OK;
}

// Object-space vertices (Scene::vertices) and the meshes' vertex ranges, kept on the device so that a transform edit can be applied there.
int fyprt_set_object_vertices(fyprt_context* c, const fyprt_vertex* object_vertices, uint32_t vertex_count, const uint32_t* mesh_first_vertex) {
    if (!c || !object_vertices || !mesh_first_vertex) return FYPRT_EINVAL;
    if (!c->haveScene) return c->fail(FYPRT_ESTATE, "fyprt_set_object_vertices before fyprt_upload_scene");
    if (c->hostOnly) return c->fail(FYPRT_ESTATE, "fyprt_set_object_vertices needs a device (host-only context)");
    if (vertex_count != c->vertexCount) return c->fail(FYPRT_EINVAL, "fyprt_set_object_vertices: vertex count differs from the uploaded scene");
    const uint32_t nM = (uint32_t)c->topoMeshes.size();
    for (uint32_t m = 0; m < nM; ++m) if (mesh_first_vertex[m] > mesh_first_vertex[m + 1] || mesh_first_vertex[m + 1] > vertex_count) return c->fail(FYPRT_EINVAL, "fyprt_set_object_vertices: mesh vertex ranges out of order / out of range");
    // every triangle of a mesh must index into that mesh's vertex range (the transform of one mesh must not move another mesh's triangles)
    for (uint32_t m = 0; m < nM; ++m)
        for (uint32_t t = c->topoMeshes[m].first_triangle; t < c->topoMeshes[m].first_triangle + c->topoMeshes[m].triangle_count; ++t)
            for (int k = 0; k < 3; ++k) { const uint32_t v = c->topoTris[(size_t)t * 4 + k]; if (v < mesh_first_vertex[m] || v >= mesh_first_vertex[m + 1]) return c->fail(FYPRT_EINVAL, "fyprt_set_object_vertices: triangle " + std::to_string(t) + " uses a vertex outside its mesh's range"); }
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c));
    HIPCHK(c, c->objVerts.alloc(vertex_count));
    if (upload(c, c->objVerts.p, object_vertices, c->objVerts.bytes())) return FYPRT_EHIP;
    c->meshFirstVertex.assign(mesh_first_vertex, mesh_first_vertex + nM + 1);
    return FYPRT_OK;
}

// A transform edit of `count` meshes (SceneManager::PerformAllSceneUpdates with meshTransformToBeUpdated, SceneManager.cpp:24-66): 64
// bytes per mesh cross the bus; the world vertices are recomputed on the device (k_transform_vertices), per-triangle records, leaf
// triangles, tree boxes and light records refreshed there, and only the moved EMISSIVE meshes' light trees are rebuilt on the host
// (their vertices read back) + the small TLAS.  Same result as fyprt_update_vertices with host-computed world vertices.
int fyprt_update_transforms(fyprt_context* c, const uint32_t* mesh_indices, const float* matrices16, uint32_t count) {
    if (!c || (count && (!mesh_indices || !matrices16))) return FYPRT_EINVAL;
    if (!c->haveScene || c->meshFirstVertex.empty()) return c->fail(FYPRT_ESTATE, "fyprt_update_transforms before fyprt_upload_scene + fyprt_set_object_vertices");
    if (c->prebuiltLightTrees) return c->fail(FYPRT_ESTATE, "fyprt_update_transforms: the scene was uploaded with prebuilt light trees; upload it again instead");
    const uint32_t nM = (uint32_t)c->topoMeshes.size(), nT = (uint32_t)(c->topoTris.size() / 4);
    for (uint32_t k = 0; k < count; ++k) if (mesh_indices[k] >= nM) return c->fail(FYPRT_EINVAL, "fyprt_update_transforms: mesh index out of range");
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c));
    std::vector<uint8_t> touched(nM, 0);
    bool lightsMoved = false;
    for (uint32_t k = 0; k < count; ++k) {
        const uint32_t m = mesh_indices[k], first = c->meshFirstVertex[m], n = c->meshFirstVertex[m + 1] - first;
        Mat4 M; std::memcpy(M.m, matrices16 + (size_t)k * 16, 64);
        if (n) hipLaunchKernelGGL(k_transform_vertices, dim3((n + 255u) / 256u), dim3(256), 0, c->stream, c->objVerts.p, c->dverts.p, first, n, M);
        const fyprt_material& mat = c->topoMats[c->topoMeshes[m].material_index];
        const float ex = mat.emission_color[0] * mat.emission_power, ey = mat.emission_color[1] * mat.emission_power, ez = mat.emission_color[2] * mat.emission_power;
        if (((ex * ex + ey * ey) + ez * ez) > 0.0f && n) {            // an emissive mesh moved: its light tree is rebuilt from its new vertices
            HIPCHK(c, hipMemcpyAsync(c->hostVerts.data() + first, c->dverts.p + first, (size_t)n * sizeof(fyprt_vertex), hipMemcpyDeviceToHost, c->stream));
            touched[m] = 1; lightsMoved = true;
        }
    }
    HIPCHK(c, hipGetLastError());
    if (nT) hipLaunchKernelGGL(k_refresh_triangles, dim3((nT + 255u) / 256u), dim3(256), 0, c->stream, c->dverts.p, c->triIdx.p, c->triPos.p, c->triShade.p, nT);
    { const int rr = run_refit(c); if (rr != FYPRT_OK) return rr; }
    c->hostBvhStale = true; c->hostVertsStale = true;
    if (lightsMoved) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const int rc = rebuild_light_trees(c, c->hostVerts.data(), touched.data());
        if (rc != FYPRT_OK) return rc;
    }
    if (c->dsc.emissiveCount) hipLaunchKernelGGL(k_build_light_records, dim3((c->dsc.emissiveCount + 255u) / 256u), dim3(256), 0, c->stream, c->dsc, c->lightRecs.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, sync_all(c));
    return FYPRT_OK;
}

int fyprt_set_camera(fyprt_context* c, const fyprt_camera_desc* cam) {
    if (!c || !cam) return FYPRT_EINVAL;
    if (cam->viewport_width == 0 || cam->viewport_height == 0) return c->fail(FYPRT_EINVAL, "fyprt_set_camera: empty viewport");
    DevCamera& d = c->dcam;
    std::memcpy(&d.invProj, cam->inverse_projection, 64); std::memcpy(&d.invView, cam->inverse_view, 64);
    float pv[16]; matmul_cm(cam->prev_projection, cam->prev_view, pv);
    std::memcpy(&d.prevProjView, pv, 64);
    d.position = f3{cam->position[0], cam->position[1], cam->position[2]};
    d.W = cam->viewport_width; d.H = cam->viewport_height;
    c->haveCamera = true;
    return FYPRT_OK;
}

// (Re)sizes the buffers of the wavefront path engine: `entries` live paths at most, `raysPer` rays per path and step,
// `stride` float4s of per-pixel state, `counters` list counters.  Grown on demand, kept between frames.
static int ensure_paths(fyprt_context* c, size_t entries, uint32_t raysPer, uint32_t stride, size_t counters) {
    const size_t npx = (size_t)c->W * c->H;
    if (c->wfRays[0].n < entries * raysPer * 3 || c->wfHits[0].n < entries * raysPer) {
        HIPCHK(c, sync_all(c));
        for (int k = 0; k < 2; ++k) { HIPCHK(c, c->wfRays[k].alloc(entries * raysPer * 3)); HIPCHK(c, c->wfHits[k].alloc(entries * raysPer)); }
    }
    if (c->wfState.n < npx * stride) { HIPCHK(c, sync_all(c)); HIPCHK(c, c->wfState.alloc(npx * stride)); }
    if (c->wfPixels.n < entries) { HIPCHK(c, sync_all(c)); HIPCHK(c, c->wfPixels.alloc(entries)); HIPCHK(c, c->wfPixels2.alloc(entries)); }
    if (c->wfCounters.n < counters) { HIPCHK(c, sync_all(c)); HIPCHK(c, c->wfCounters.alloc(counters)); }
    return FYPRT_OK;
}

typedef void (*shade_kernel_t)(DevScene, DevCamera, DevFrame, DevSettings, PathIO);
static shade_kernel_t shade_kernel(int stage) {
    switch (stage) {
        case T_BRUTE: return k_shade<T_BRUTE>; case T_UNIFORM: return k_shade<T_UNIFORM>; case T_COSINE: return k_shade<T_COSINE>;
        case T_GGX: return k_shade<T_GGX>; case T_BRDF: return k_shade<T_BRDF>; case T_LIGHT: return k_shade<T_LIGHT>;
        case T_NEE: return k_shade<T_NEE>; case T_GI1: return k_shade<T_GI1>; default: return k_shade<T_GI2>;
    }
}

// phase: 0 = the whole frame; 1 = ReSTIR Part 1 only (nothing of the frame's bookkeeping advances); 2 = the rest of the frame that a
// phase-1 call started.  The split exists for the halo EXCHANGE of a multi-GPU frame (fyprt_multi.h): Part 1 on every band, the
// bands' Part-1 records of each other's halo rows copied across, Part 2 on every band.
// The "previous normals" of a ReSTIR frame are the last ReSTIR frame's, whichever of the two techniques rendered it (the reference keeps
// one pair of normal buffers for both; rt_refit.h: k_sync_history_normals).  Called before Part 1 — by the multi-GPU layer before the
// halo rows' history is fetched from their owners, so that what travels is already in step.
static int sync_restir_normals(fyprt_context* c, int tech, hipStream_t stream) {
    if (c->lastRestir < 0 || c->lastRestir == tech || c->W == 0) { return FYPRT_OK; }
    const uint32_t npx = c->W * c->H;
    DIRec* records = c->dprevFlip ? c->dprevB.p : c->dprevA.p;            // the history the next ReSTIR DI frame reads
    f2* normals = c->normalFlip ? c->normalB.p : c->normalA.p;            // the previous normals the next ReSTIR GI frame reads
    hipLaunchKernelGGL(k_sync_history_normals, dim3((npx + 255u) / 256u), dim3(256), 0, stream, records, normals, npx, tech == FYPRT_RESTIR_DI ? 1 : 0);
    c->lastRestir = tech;
    return c->hip(hipGetLastError(), "k_sync_history_normals");
}

static int enqueue_frame_impl(fyprt_context* c, const fyprt_settings* s, bool timed, int phase);
static int enqueue_frame(fyprt_context* c, const fyprt_settings* s, bool timed, int phase = 0) {
    const int rc = enqueue_frame_impl(c, s, timed, phase);
    if (rc != FYPRT_OK && rc != FYPRT_ESTATE) c->part1Pending = false;      // a frame that failed half-way is abandoned, not left pending
    return rc;
}
static int enqueue_frame_impl(fyprt_context* c, const fyprt_settings* s, bool timed, int phase) {
    if (c->hostOnly) return c->fail(FYPRT_ESTATE, "host-only context (device -1) cannot render");
    if (!c->haveScene || !c->haveCamera || c->W == 0) return c->fail(FYPRT_ESTATE, "fyprt_render: resize, upload_scene and set_camera must precede render");
    if (c->dcam.W != c->W || c->dcam.H != c->H) return c->fail(FYPRT_ESTATE, "fyprt_render: camera viewport differs from the render size");
    const int tech = s->technique;
    if (tech < 0 || tech > 8) return c->fail(FYPRT_EINVAL, "fyprt_render: unknown technique");
    if (phase != 0 && tech != FYPRT_RESTIR_DI && tech != FYPRT_RESTIR_GI) return c->fail(FYPRT_EINVAL, "fyprt_render_part: only the ReSTIR techniques have two parts");
    if (phase == 2 && !c->part1Pending) return c->fail(FYPRT_ESTATE, "fyprt_render_part(2) without a preceding part 1");
    if (phase != 2 && c->part1Pending) return c->fail(FYPRT_ESTATE, "a frame's part 1 is pending: call fyprt_render_part(ctx, settings, 2) first");
    if ((tech == FYPRT_LIGHT_SOURCE_SAMPLING || tech == FYPRT_NEE) && (c->dsc.emissiveCount == 0 || c->dsc.ltTlasCount == 0))
        return c->fail(FYPRT_ENOLIGHT, "fyprt_render: technique needs emissive triangles and a light tree");
    if (tech == FYPRT_RESTIR_DI && c->dsc.emissiveCount == 0) return c->fail(FYPRT_ENOLIGHT, "fyprt_render: ReSTIR DI needs emissive triangles");
    HIPCHK(c, hipSetDevice(c->device));
    DevSettings st;
    st.sky = f3{s->sky_color[0], s->sky_color[1], s->sky_color[2]};
    st.maxBounces = (uint8_t)s->light_bounces; st.sampleCount = (uint8_t)s->sample_count;       // Renderer.cu:2444, :2480-2481
    st.candidateCount = (uint32_t)s->light_candidate_count; st.randSeed = s->rand_seed;
    st.useTemporal = s->use_temporal_reuse ? 1u : 0u; st.useSpatial = s->use_spatial_reuse ? 1u : 0u;
    st.historyLimit = (uint8_t)s->temporal_history_limit; st.numNeighbors = (uint8_t)s->spatial_neighbor_num; st.radius = (uint8_t)s->spatial_neighbor_radius;
    st.skipDeadRays = c->tuning[18] ? 1u : 0u;
    DevFrame fr;
    fr.accum = c->accum.p; fr.image = c->externalImage ? c->externalImage : c->image.p; fr.payload = c->payload.p; fr.depth = c->depth.p;
    fr.normalPrev = c->normalFlip ? c->normalB.p : c->normalA.p; fr.normalCur = c->normalFlip ? c->normalA.p : c->normalB.p;
    fr.di = c->di.p; fr.diPrev = c->diPrev.p; fr.gi = c->gi.p; fr.giPrev = c->giPrev.p; fr.giHot = c->giHot.p;
    fr.drec = c->drec.p; fr.dprevRead = c->dprevFlip ? c->dprevB.p : c->dprevA.p; fr.dprevWrite = c->dprevFlip ? c->dprevA.p : c->dprevB.p;
    fr.W = c->W; fr.H = c->H; fr.frameIndex = c->frameIndex; fr.rowBegin = c->rowBegin; fr.rowEnd = c->rowEnd;
    fr.stripeRows = 0; fr.stripeParts = 1; fr.stripePart = 0;
    const bool striped = c->stripeRows != 0 && c->stripeRows < c->H && c->stripeParts > 1;
    if (striped && (tech == FYPRT_RESTIR_DI || tech == FYPRT_RESTIR_GI)) return c->fail(FYPRT_ESTATE, "ReSTIR frames need contiguous rows: clear fyprt_set_row_stripes first");
    fr.histBegin = (tech == FYPRT_RESTIR_GI) ? c->histGI[0] : c->histDI[0]; fr.histEnd = (tech == FYPRT_RESTIR_GI) ? c->histGI[1] : c->histDI[1];
    c->dsc.rayCounter = c->countRays ? c->rayCounter.p : nullptr;
    // node-loop quorum of the fused per-pixel kernels (key 7): 0 = auto — 16 for the light-tree kernels (their shadow rays: NEE 5.2 -> 4.95 ms),
    // none elsewhere (path and ReSTIR GI kernels: neutral or slightly negative)
    c->dsc.nodeQuorum = (uint32_t)c->tuning[7];
    // traversal-stack budget (node_step's rule): never below the level count (the induction), never above the 31 the node
    // format records; by default a few entries above the level count, so the LDS stack is no larger than this tree needs
    // and more workgroups fit a CU (LDS is what limits residency: (budget + 1) KB per 256-thread workgroup)
    const int budget = effective_stack_budget(c);
    c->dsc.stackBudget = budget;
#ifdef RT_TOPCACHE
    c->dsc.topCount = (uint32_t)std::min<size_t>((size_t)std::max(0, c->tuning[16]), c->hostBvh.nodes.size());
#else
    c->dsc.topCount = 0u;                      // tuning key 16 only acts in a -DRT_TOPCACHE build (rt_device.h: measured slower)
#endif
    const size_t ldsBytes = (size_t)(budget + 1) * kBlock * sizeof(int32_t) + (size_t)c->dsc.topCount * 64u;
    // Pipelining (tuning key 11): a wavefront ReSTIR DI frame runs Part 1 + setup on the front stream and the trace kernel on
    // `stream`.  Nothing the front part writes is read or written by a trace kernel (payload, records, history, depth, its own
    // task queue — two queues alternate), and image + accumulation are touched by trace kernels only (p1Mode 1), which stay in
    // frame order on `stream`; so frame N+1's front part may run beside frame N's trace kernel.  Any other frame runs on
    // `stream` alone, after everything before it.
    const int par = (int)(c->frameSerial & 1ull);
    const bool wavefront = tech == FYPRT_RESTIR_DI && c->tuning[1] == 1;
    const bool overlap = wavefront && c->tuning[11] != 0 && !c->countRays && phase == 0;
    hipStream_t fs = overlap ? c->front : c->stream;             // where Part 1 + setup go
    if (overlap) {
        HIPCHK(c, hipStreamWaitEvent(c->front, c->evDone[par], 0));                            // frame N-2 done: its queue is free
        if (!c->lastOverlapped) HIPCHK(c, hipStreamWaitEvent(c->front, c->evDone[par ^ 1], 0));   // frame N-1 ran on `stream` alone
    }
    if (c->countRays && phase != 2) HIPCHK(c, hipMemsetAsync(c->rayCounter.p, 0, 256, c->stream));
    // frame 1 (or toAccumulate == false): the accumulator starts from zero (Renderer.cu:50-51) — on `stream`, which owns it
    // (the whole buffer, as the reference does, not just this context's rows: a band moved later with fyprt_set_rows must not find the
    // sums of an earlier accumulation in its new rows)
    if (c->frameIndex == 1 && phase != 2) HIPCHK(c, hipMemsetAsync(c->accum.p, 0, c->accum.bytes(), c->stream));
    const uint32_t tilesX = (c->W + 15u) / 16u;
    fr.tileOrder = (uint32_t)c->tuning[0];
    fr.p1Mode = wavefront ? 1u : 0u;
    auto gridFor = [&](uint32_t rb, uint32_t re) {
        const uint32_t tilesY = (re - rb + 15u) / 16u;
        if (c->tuning[0] == 2) return dim3(tilesX * ((tilesY + 7u) / 8u) * 8u);
        return dim3(((tilesX * tilesY + 7u) / 8u) * 8u);
    };
    const dim3 block(kBlock);
    const dim3 grid = gridFor(c->rowBegin, c->rowEnd);
    int ei = 0;
    c->ev = c->ring[c->frameSerial % fyprt_context::kRing];
    if (phase == 2) ei = 1;
    else if (timed) HIPCHK(c, hipEventRecord(c->ev[ei++], fs));
    int launches = 0;
    // ---- wavefront path engine (rt_paths.h): primary kernel, then per step one shade launch + one persistent trace launch
    struct StageRun { int stage; uint32_t steps, raysPer, stride; const uint32_t* pixelList; uint32_t* cnt; uint32_t* heads; uint32_t* part2List; uint32_t* part2Count; int counterPart; uint32_t* misCounts; size_t maxEntries; };
    auto run_stage = [&](const StageRun& r) -> int {
        const shade_kernel_t shade = shade_kernel(r.stage);
        const dim3 shadeGrid((uint32_t)(c->numCUs * 8));
        if (c->pathOccLds != ldsBytes) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_rays<false>, kBlock, ldsBytes) != hipSuccess || n <= 0) n = 4;
            c->pathOcc = n; c->pathOccLds = ldsBytes;
        }
        const int perCU = c->tuning[2] > 0 ? c->tuning[2] : c->pathOcc;
        DevScene tsc = c->dsc;
        tsc.nodeQuorum = (uint32_t)c->tuning[6];                 // incoherent rays: leave the node loop once few lanes remain in it
        tsc.rayCounter = c->countRays ? c->rayCounter.p + 8 * r.counterPart : nullptr;
        for (uint32_t it = 0; it <= r.steps; ++it) {
            PathIO io{};
            io.fusedOwner = kNotFused;
            io.pixelList = r.pixelList; io.raysIn = c->wfRays[it & 1u].p; io.hitsIn = c->wfHits[it & 1u].p; io.countIn = r.cnt + it;
            io.raysOut = c->wfRays[(it + 1u) & 1u].p; io.countOut = r.cnt + it + 1; io.state = c->wfState.p; io.stateStride = r.stride;
            io.iteration = it; io.raysPer = r.raysPer; io.part2List = r.part2List; io.part2Count = r.part2Count;
            // NEE's MIS list reuses the primary kernel's pixel list, which only step 0 reads (and step 0 has no ray results, hence no MIS entries)
            io.misList = c->wfPixels.p; io.misCount = r.misCounts ? r.misCounts + it : nullptr;
            if (r.stage == T_GI2) {                                  // owner lists ping-pong between the Part-2 list's buffer and the (by now free) primary list's
                io.ownersIn = (it & 1u) ? c->wfPixels.p : c->wfPixels2.p; io.ownersOut = (it & 1u) ? c->wfPixels2.p : c->wfPixels.p;
            }
            hipLaunchKernelGGL(shade, shadeGrid, block, 0, c->stream, c->dsc, c->dcam, fr, st, io);
            if (r.stage == T_NEE && it > 0u)                         // NEE: emitter-hit MIS for the few paths that need it (may add to the pick list)
                hipLaunchKernelGGL(k_nee_mis, dim3((uint32_t)c->numCUs), block, 0, c->stream, c->dsc, c->dcam, fr, st, (const uint32_t*)io.misList, (const uint32_t*)io.misCount,
                                   c->wfState.p, r.stride, r.part2List, io.countOut);
            if (it == r.steps) break;                                // the last step only consumes: every path has emitted all its rays
            if (r.stage == T_NEE)                                    // light pick + ray construction for the listed paths
                hipLaunchKernelGGL(k_nee_emit, shadeGrid, block, 0, c->stream, c->dsc, st, (const uint32_t*)r.part2List, (const uint32_t*)io.countOut, c->wfState.p, r.stride, io.raysOut, r.raysPer);
            TraceQueue q{};
            q.rays = io.raysOut; q.hits = c->wfHits[(it + 1u) & 1u].p; q.count = io.countOut; q.raysPer = r.raysPer; q.head = r.heads + it + 1;
            q.chunk = (uint32_t)(c->tuning[4] > 0 ? c->tuning[4] : 128); q.refillLanes = (uint32_t)(c->tuning[5] > 0 ? c->tuning[5] : 24);
            q.staticChunks = (uint32_t)(c->tuning[9] > 0 ? c->tuning[9] : 2); q.minChunk = (uint32_t)(c->tuning[10] > 0 ? c->tuning[10] : q.chunk);      // (auto: two static chunks per wave — config 3 3.13 -> 3.08 ms, profiles/README.md r03)
            // small trees (cheap rays): one thread per ray; big ones: persistent waves with lane refill (tuning key 15: 0 = by tree size)
            const bool simple = c->tuning[15] == 2 || (c->tuning[15] == 0 && c->hostBvh.tris.size() < 65536u);
            if (simple) {
                const uint32_t sg = (uint32_t)std::min<size_t>((size_t)c->numCUs * 16u, (r.maxEntries * r.raysPer + kBlock - 1) / kBlock);
                if (c->countRays) hipLaunchKernelGGL(k_trace_rays_simple<true>, dim3(std::max(1u, sg)), block, ldsBytes, c->stream, tsc, q);
                else hipLaunchKernelGGL(k_trace_rays_simple<false>, dim3(std::max(1u, sg)), block, ldsBytes, c->stream, tsc, q);
            }
            else if (c->countRays) hipLaunchKernelGGL(k_trace_rays<true>, dim3((uint32_t)(c->numCUs * perCU)), block, ldsBytes, c->stream, tsc, q);
            else hipLaunchKernelGGL(k_trace_rays<false>, dim3((uint32_t)(c->numCUs * perCU)), block, ldsBytes, c->stream, tsc, q);
            // long sample x bounce products: stop once no path is alive any more.  Only inside the blocking fyprt_render — an asynchronous
            // call (fyprt_render_async, group / comm frames) must not wait on the device: there the remaining steps are launched and find
            // empty lists (every kernel of a step returns at once on a count of zero)
            if (c->blockingCall && r.steps > 8u && (it & 3u) == 3u) {
                uint32_t alive = 0;
                HIPCHK(c, hipMemcpyAsync(&alive, io.countOut, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (alive == 0u) break;
            }
        }
        return c->hip(hipGetLastError(), "path stage launch");
    };
    switch (tech) {
        case FYPRT_BRUTE_FORCE: case FYPRT_UNIFORM_SAMPLING: case FYPRT_COSINE_WEIGHTED_SAMPLING: case FYPRT_GGX_SAMPLING: case FYPRT_BRDF_SAMPLING:
        case FYPRT_LIGHT_SOURCE_SAMPLING: case FYPRT_NEE: {
            // rays a pixel emits one after the other = trace passes of the frame
            const uint32_t nSamples = (tech == FYPRT_BRUTE_FORCE) ? 1u : st.sampleCount;
            const uint32_t steps = (tech == FYPRT_LIGHT_SOURCE_SAMPLING) ? nSamples : nSamples * st.maxBounces;
            const uint32_t raysPer = (tech == FYPRT_NEE && st.maxBounces != 1u) ? 2u : 1u;
            dim3 pgrid = grid;
            if (striped) {                                           // k_primary maps the local rows [0, n) onto this context's stripes
                fr.stripeRows = c->stripeRows; fr.stripeParts = c->stripeParts; fr.stripePart = c->stripePart;
                fr.rowBegin = 0; fr.rowEnd = stripe_row_count(c->H, c->stripeRows, c->stripeParts, c->stripePart);
                pgrid = gridFor(fr.rowBegin, fr.rowEnd);
            }
            const size_t entries = (size_t)(fr.rowEnd - fr.rowBegin) * c->W, L = (size_t)steps + 2;
            const uint32_t stride = (tech == FYPRT_NEE) ? 6u : 2u;
            { const int rc = ensure_paths(c, entries, raysPer, stride, 3 * L); if (rc != FYPRT_OK) return rc; }
            HIPCHK(c, hipMemsetAsync(c->wfCounters.p, 0, 3 * L * sizeof(uint32_t), c->stream));
            c->dsc.nodeQuorum = (uint32_t)c->tuning[7];             // coherent primary rays
            // small trees, techniques 0-5: the whole frame in one launch, one thread per pixel (rt_paths.h: k_path_fused; key 17: 0 = by tree size, 1 = never, 2 = always)
            const bool fused = tech != FYPRT_NEE && (c->tuning[17] == 2 || (c->tuning[17] == 0 && c->hostBvh.tris.size() < 65536u));
            if (fused) {
                PathIO io{};
                io.fusedOwner = kNotFused; io.raysIn = c->wfRays[0].p; io.raysOut = c->wfRays[0].p; io.hitsIn = c->wfHits[0].p; io.state = c->wfState.p; io.stateStride = stride; io.raysPer = 1u;
                typedef void (*fused_kernel_t)(DevScene, DevCamera, DevFrame, DevSettings, PathIO, uint32_t);
                static const fused_kernel_t kFused[2][6] = {{k_path_fused<T_BRUTE, false>, k_path_fused<T_UNIFORM, false>, k_path_fused<T_COSINE, false>, k_path_fused<T_GGX, false>, k_path_fused<T_BRDF, false>, k_path_fused<T_LIGHT, false>},
                                                            {k_path_fused<T_BRUTE, true>, k_path_fused<T_UNIFORM, true>, k_path_fused<T_COSINE, true>, k_path_fused<T_GGX, true>, k_path_fused<T_BRDF, true>, k_path_fused<T_LIGHT, true>}};
                hipLaunchKernelGGL(kFused[c->countRays ? 1 : 0][tech], pgrid, block, ldsBytes, c->stream, c->dsc, c->dcam, fr, st, io, steps);
                launches = 1;
                break;
            }
            if (c->countRays) hipLaunchKernelGGL(k_primary<true>, pgrid, block, ldsBytes, c->stream, c->dsc, c->dcam, fr, st, c->wfPixels.p, c->wfCounters.p);
            else hipLaunchKernelGGL(k_primary<false>, pgrid, block, ldsBytes, c->stream, c->dsc, c->dcam, fr, st, c->wfPixels.p, c->wfCounters.p);
            StageRun r{tech, steps, raysPer, stride, c->wfPixels.p, c->wfCounters.p, c->wfCounters.p + L, (tech == FYPRT_NEE) ? c->wfPixels2.p : nullptr, nullptr, 0, (tech == FYPRT_NEE) ? c->wfCounters.p + 2 * L : nullptr, entries};
            { const int rc = run_stage(r); if (rc != FYPRT_OK) return rc; }
            launches = 1;
            break;
        }
        case FYPRT_RESTIR_DI: case FYPRT_RESTIR_GI: {
            // halo rows: Part 1 recomputed on them (default), or — exchange mode — left to the band that owns them and copied in between the parts
            const uint32_t p1halo = (c->haloExchange || c->tuning[13]) ? 0u : c->halo;
            const uint32_t p1b = (c->rowBegin > p1halo) ? c->rowBegin - p1halo : 0u;
            const uint32_t p1e = (c->rowEnd + p1halo < c->H) ? c->rowEnd + p1halo : c->H;
            c->dsc.nodeQuorum = (uint32_t)c->tuning[7];             // Part 1 traces coherent primary rays only
            // The reference's spatial-neighbour coordinate is computed in unsigned arithmetic (R.cu:1916-1917): an offset
            // above the first row wraps and clamps to the LAST row.  A band that owns rows < radius therefore also needs
            // Part 1 of row H-1 (one extra row of recompute) to stay bit-identical to a single-GPU frame.  It rides in the same
            // launch as one more row of tiles (a separate one-row launch is all latency: ~0.09 ms on a 135-row band).
            const bool extra = p1halo > 0 && c->rowBegin < p1halo && p1e < c->H;
            const uint32_t extraRow = extra ? c->H - 1u : 0xFFFFFFFFu;
            const dim3 g1 = gridFor(p1b, extra ? p1e + 16u : p1e);
            if (tech == FYPRT_RESTIR_GI) {
                // Part 1 = primary kernel + bounce-loop steps (they build the Part-2 list as paths complete); Part 2 = neighbour-loop steps
                const size_t p1px = ((size_t)(p1e - p1b) + (extra ? 1u : 0u)) * c->W;
                const uint32_t steps1 = st.maxBounces, steps2 = st.useSpatial ? st.numNeighbors : 0u;
                const size_t L1 = (size_t)steps1 + 2, L2 = (size_t)steps2 + 2;
                { const int rc = ensure_paths(c, p1px, 1, 5, 2 * L1 + 2 * L2); if (rc != FYPRT_OK) return rc; }
                uint32_t* cnt1 = c->wfCounters.p; uint32_t* cnt2 = cnt1 + 2 * L1;      // cnt2[0] = length of the Part-2 list
                if (phase != 2) { const int rc = sync_restir_normals(c, tech, c->stream); if (rc != FYPRT_OK) return rc; }   // the last ReSTIR frame was a DI frame: its normals (in the history records) are this frame's "previous normals"
                if (phase != 2) {
                    HIPCHK(c, hipMemsetAsync(cnt1, 0, (2 * L1 + 2 * L2) * sizeof(uint32_t), c->stream));
                    if (c->countRays) hipLaunchKernelGGL(k_gi_primary<true>, g1, block, ldsBytes, c->stream, c->dsc, c->dcam, fr, st, p1b, p1e, extraRow, c->wfPixels.p, cnt1);
                    else hipLaunchKernelGGL(k_gi_primary<false>, g1, block, ldsBytes, c->stream, c->dsc, c->dcam, fr, st, p1b, p1e, extraRow, c->wfPixels.p, cnt1);
                    StageRun r1{T_GI1, steps1, 1u, 5u, c->wfPixels.p, cnt1, cnt1 + L1, c->wfPixels2.p, cnt2, 0, nullptr, p1px};
                    { const int rc = run_stage(r1); if (rc != FYPRT_OK) return rc; }
                    if (timed) HIPCHK(c, hipEventRecord(c->ev[ei++], c->stream));
                    if (phase == 1) { c->part1Pending = true; return c->hip(hipGetLastError(), "ReSTIR GI part 1"); }
                }
                if (c->tuning[19] == 2) {
                    // Part 2 as one PERSISTENT launch (rt_paths.h: k_gi2_persistent): a lane owns a pixel of the list, lanes without a ray in flight are serviced together
                    DevScene tsc = c->dsc;
                    tsc.nodeQuorum = (uint32_t)c->tuning[6];
                    tsc.rayCounter = c->countRays ? c->rayCounter.p + 8 : nullptr;
                    if (c->gi2OccLds != ldsBytes) {
                        int nb = 0;
                        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_gi2_persistent<false>, kBlock, ldsBytes) != hipSuccess || nb <= 0) nb = 3;
                        c->gi2Occ = nb; c->gi2OccLds = ldsBytes;
                    }
                    const int perCU = c->tuning[2] > 0 ? std::min(c->tuning[2], c->gi2Occ) : c->gi2Occ;
                    GI2Queue gq{};
                    gq.list = c->wfPixels2.p; gq.count = cnt2; gq.head = cnt2 + L2 + 1;
                    gq.chunk = (uint32_t)(c->tuning[4] > 0 ? c->tuning[4] : 128); gq.refillLanes = (uint32_t)(c->tuning[20] > 0 ? c->tuning[20] : 48);
                    gq.staticChunks = (uint32_t)(c->tuning[9] > 0 ? c->tuning[9] : 2); gq.minChunk = (uint32_t)(c->tuning[10] > 0 ? c->tuning[10] : gq.chunk);
                    if (c->countRays) hipLaunchKernelGGL(k_gi2_persistent<true>, dim3((uint32_t)(c->numCUs * perCU)), block, ldsBytes, c->stream, tsc, c->dcam, fr, st, gq);
                    else hipLaunchKernelGGL(k_gi2_persistent<false>, dim3((uint32_t)(c->numCUs * perCU)), block, ldsBytes, c->stream, tsc, c->dcam, fr, st, gq);
                    HIPCHK(c, hipGetLastError());
                } else if (c->tuning[19] == 1) {
                    // Part 2 in one launch (rt_paths.h: k_gi2_fused): one thread per listed pixel for the whole neighbour loop, visibility rays traced in place
                    DevScene tsc = c->dsc;
                    tsc.nodeQuorum = (uint32_t)c->tuning[6];
                    tsc.rayCounter = c->countRays ? c->rayCounter.p + 8 : nullptr;
                    const uint32_t fg = (uint32_t)std::max<size_t>(1, std::min<size_t>((size_t)c->numCUs * 64u, (p1px + kBlock - 1) / kBlock));
                    if (c->countRays) hipLaunchKernelGGL(k_gi2_fused<true>, dim3(fg), block, ldsBytes, c->stream, tsc, c->dcam, fr, st, (const uint32_t*)c->wfPixels2.p, (const uint32_t*)cnt2);
                    else hipLaunchKernelGGL(k_gi2_fused<false>, dim3(fg), block, ldsBytes, c->stream, tsc, c->dcam, fr, st, (const uint32_t*)c->wfPixels2.p, (const uint32_t*)cnt2);
                    HIPCHK(c, hipGetLastError());
                } else {
                    StageRun r2{T_GI2, steps2, 1u, 5u, c->wfPixels2.p, cnt2, cnt2 + L2, nullptr, nullptr, 1, nullptr, p1px};
                    const int rc = run_stage(r2); if (rc != FYPRT_OK) return rc;
                }
                launches = 2;
                c->normalFlip = !c->normalFlip; c->histGI[0] = c->rowBegin; c->histGI[1] = c->rowEnd;
                break;
            }
            if (phase != 2) { const int rc = sync_restir_normals(c, tech, fs); if (rc != FYPRT_OK) return rc; }   // the last ReSTIR frame was a GI frame: its normals are this frame's "previous normals"
            if (phase != 2) {
                if (c->countRays) hipLaunchKernelGGL(k_di_part1<true>, g1, block, ldsBytes, fs, c->dsc, c->dcam, fr, st, p1b, p1e, extraRow);
                else hipLaunchKernelGGL(k_di_part1<false>, g1, block, ldsBytes, fs, c->dsc, c->dcam, fr, st, p1b, p1e, extraRow);
                if (timed) HIPCHK(c, hipEventRecord(c->ev[ei++], fs));
                if (phase == 1) { c->part1Pending = true; return c->hip(hipGetLastError(), "ReSTIR DI part 1"); }
            }
            c->dsc.nodeQuorum = (uint32_t)c->tuning[6];   // shadow-ray kernels of ReSTIR DI Part 2: measured 0.85 -> 0.68 ms
            if (c->countRays) c->dsc.rayCounter = c->rayCounter.p + 8;      // per-launch counters
            launches = 2;
            if (tech == FYPRT_RESTIR_DI && c->tuning[1] == 1) {
                ShadowQueue q{};
                q.tasks = c->shadowTasks.p + (size_t)par * c->queueStride; q.counters = c->queueCounters.p + 4 * par;
                q.chunk = (uint32_t)(c->tuning[4] > 0 ? c->tuning[4] : 128); q.refillLanes = (uint32_t)(c->tuning[5] > 0 ? c->tuning[5] : 24); q.staticChunks = (uint32_t)(c->tuning[9] > 0 ? c->tuning[9] : 2); q.minChunk = (uint32_t)(c->tuning[10] > 0 ? c->tuning[10] : q.chunk);
                const size_t sg = (size_t)par * c->sortGroups;
                q.sortMode = c->tuning[3] ? 1u : 0u; q.numGroups = grid.x; q.counts = c->sortCounts.p + sg; q.keys = c->sortKeys.p + sg * 256u; q.hist = c->sortHist.p + sg * kSortBins;
                q.binOffset = c->sortOffset.p + sg * kSortBins; q.binTotal = c->sortTotal.p + (size_t)par * kSortBins; q.sorted = c->sortIndex.p + sg * 256u;
                HIPCHK(c, hipMemsetAsync(q.counters, 0, 16, fs));
                if (c->tuning[14] == 1) hipLaunchKernelGGL(k_di_part2_setup<1>, grid, block, 0, fs, c->dsc, c->dcam, fr, st, q);
                else if (c->tuning[14] == 2) hipLaunchKernelGGL(k_di_part2_setup<2>, grid, block, 0, fs, c->dsc, c->dcam, fr, st, q);
                else hipLaunchKernelGGL(k_di_part2_setup<0>, grid, block, 0, fs, c->dsc, c->dcam, fr, st, q);
                if (q.sortMode) {
                    hipLaunchKernelGGL(k_di_sort_scan, dim3(kSortBins), block, 0, fs, q);
                    hipLaunchKernelGGL(k_di_sort_scatter, grid, block, 0, fs, q);
                }
                if (timed) HIPCHK(c, hipEventRecord(c->ev[ei++], fs));
                if (overlap) {                                   // the trace kernel waits for this frame's front part only
                    HIPCHK(c, hipEventRecord(c->evFront[par], c->front));
                    HIPCHK(c, hipStreamWaitEvent(c->stream, c->evFront[par], 0));
                    if (timed) HIPCHK(c, hipEventRecord(c->ev[4], c->stream));      // start of the trace kernel on its own stream
                }
                if (c->countRays) c->dsc.rayCounter = c->rayCounter.p + 16;
                int perCU = c->tuning[2];
                if (perCU <= 0) {          // as many workgroups as registers + LDS let a CU hold (asked from the runtime once per stack size)
                    if (c->traceOccLds != ldsBytes) {
                        int n = 0;
                        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_di_part2_trace<false>, kBlock, ldsBytes) != hipSuccess || n <= 0) n = 4;
                        c->traceOcc = n; c->traceOccLds = ldsBytes;
                    }
                    perCU = c->traceOcc;
                    // pipelined frames: the persistent grid shares the chip with Part 1 + setup of the NEXT frame; with every slot a CU has
                    // (6 workgroups) those start late and run in extra rounds — 4 per CU leave them room: an eighth of the frame (a multi-GPU
                    // band) renders in 0.228 instead of 0.252 ms, the whole frame in the same 0.888 ms (profiles/README.md r03)
                    if (overlap && perCU > 4) perCU = 4;
                }
                // persistent grid: as many workgroups as the chip holds — but not more than the band has tasks for (one lane per task): a narrow
                // multi-GPU band would otherwise park idle workgroups on the LDS / wave slots the next frame's Part 1 is waiting for
                const uint32_t traceGrid = std::max(8u, std::min((uint32_t)(c->numCUs * perCU), grid.x));
                if (c->countRays) hipLaunchKernelGGL(k_di_part2_trace<true>, dim3(traceGrid), block, ldsBytes, c->stream, c->dsc, fr, q);
                else hipLaunchKernelGGL(k_di_part2_trace<false>, dim3(traceGrid), block, ldsBytes, c->stream, c->dsc, fr, q);
                launches = 3;
            }
            else if (c->countRays) hipLaunchKernelGGL(k_di_part2<true>, grid, block, ldsBytes, c->stream, c->dsc, c->dcam, fr, st);
            else hipLaunchKernelGGL(k_di_part2<false>, grid, block, ldsBytes, c->stream, c->dsc, c->dcam, fr, st);
            c->dprevFlip = !c->dprevFlip; c->histDI[0] = c->rowBegin; c->histDI[1] = c->rowEnd;
            break;
        }
    }
    HIPCHK(c, hipGetLastError());
    c->part1Pending = false;
    if (timed) HIPCHK(c, hipEventRecord(c->ev[ei++], c->stream));
    HIPCHK(c, hipEventRecord(c->evDone[par], c->stream));                // the frame is complete (and its task queue free again)
    c->lastOverlapped = overlap;
    c->lastLaunches = launches; c->lastTech = tech;
    if (tech == FYPRT_RESTIR_DI || tech == FYPRT_RESTIR_GI) c->lastRestir = tech;
    c->ringLaunches[c->frameSerial % fyprt_context::kRing] = timed ? launches : 0;
    c->ringSplit[c->frameSerial % fyprt_context::kRing] = overlap;
    c->frameSerial++;
    if (s->to_accumulate) c->frameIndex++; else c->frameIndex = 1;       // Renderer.cu:258-261
    return FYPRT_OK;
}

int fyprt_render(fyprt_context* c, const fyprt_settings* s, fyprt_frame_stats* stats) {
    if (!c || !s) return FYPRT_EINVAL;
    c->blockingCall = true;
    int rc = enqueue_frame(c, s, true);
    c->blockingCall = false;
    if (rc != FYPRT_OK) return rc;
    HIPCHK(c, sync_all(c));                            // cudaDeviceSynchronize, Renderer.cu:237
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->launches = (uint32_t)c->lastLaunches;
        float total = 0.0f;
        const bool split = c->ringSplit[(c->frameSerial - 1ull) % fyprt_context::kRing];
        for (int k = 0; k < c->lastLaunches; ++k) {
            float ms = 0.0f; (void)hipEventElapsedTime(&ms, (split && k == 2) ? c->ev[4] : c->ev[k], c->ev[k + 1]);   // split frame: the trace kernel has its own start event
            stats->kernel_ms_part[k] = ms; total += ms;
        }
        stats->kernel_ms = total;
        if (c->countRays) {
            unsigned long long r[32] = {0}; (void)hipMemcpy(r, c->rayCounter.p, 256, hipMemcpyDeviceToHost);
            for (int k = 0; k < 4; ++k) {
                stats->part_rays[k] = r[8 * k + 0]; stats->part_box_tests[k] = r[8 * k + 1]; stats->part_tri_tests[k] = r[8 * k + 2]; stats->part_hits[k] = r[8 * k + 3];
                stats->part_node_visits[k] = r[8 * k + 4];
                stats->rays += r[8 * k + 0]; stats->box_tests += r[8 * k + 1]; stats->tri_tests += r[8 * k + 2]; stats->hits += r[8 * k + 3]; stats->node_visits += r[8 * k + 4];
            }
        }
    }
    return FYPRT_OK;
}

int fyprt_render_async(fyprt_context* c, const fyprt_settings* s) { if (!c || !s) return FYPRT_EINVAL; return enqueue_frame(c, s, true); }

int fyprt_frame_timings(fyprt_context* c, uint32_t frames_back, float* kernel_ms_part4, uint32_t* launches) {
    if (!c || !kernel_ms_part4) return FYPRT_EINVAL;
    if (frames_back >= (uint32_t)fyprt_context::kRing || frames_back >= c->frameSerial) return c->fail(FYPRT_EINVAL, "fyprt_frame_timings: frame no longer in the ring");
    const unsigned long long slot = (c->frameSerial - 1ull - frames_back) % fyprt_context::kRing;
    const int n = c->ringLaunches[slot];
    for (int k = 0; k < 4; ++k) kernel_ms_part4[k] = 0.0f;
    for (int k = 0; k < n && k < 4; ++k) {
        float ms = 0.0f;
        hipError_t e = hipEventElapsedTime(&ms, (c->ringSplit[slot] && k == 2) ? c->ring[slot][4] : c->ring[slot][k], c->ring[slot][k + 1]);
        if (e != hipSuccess) return c->hip(e, "hipEventElapsedTime (synchronize the context first)");
        kernel_ms_part4[k] = ms;
    }
    if (launches) *launches = (uint32_t)n;
    return FYPRT_OK;
}
int fyprt_synchronize(fyprt_context* c) { if (!c) return FYPRT_EINVAL; if (c->hostOnly) return FYPRT_OK; HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c)); return FYPRT_OK; }

int fyprt_readback(fyprt_context* c, uint32_t* rgba8, float* accum4) {
    if (!c) return FYPRT_EINVAL;
    if (c->hostOnly || c->W == 0) return c->fail(FYPRT_ESTATE, "fyprt_readback before fyprt_resize");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    const size_t off = (size_t)c->rowBegin * c->W, cnt = (size_t)(c->rowEnd - c->rowBegin) * c->W;
    const uint32_t* img = c->externalImage ? c->externalImage : c->image.p;
    if (rgba8) HIPCHK(c, hipMemcpy(rgba8 + off, img + off, cnt * 4, hipMemcpyDeviceToHost));
    if (accum4) HIPCHK(c, hipMemcpy(accum4 + off * 4, c->accum.p + off, cnt * 16, hipMemcpyDeviceToHost));
    return FYPRT_OK;
}

// The lean correctly rounded sqrt / reciprocal / reciprocal square root of rt_math.h against the compiler's sequences on all 2^32
// arguments each (a few milliseconds).  mismatches[3] / first_bad[3] in the order sqrt, 1/x, 1/sqrt(x); all zero on a sound build.
int fyprt_selftest_math(fyprt_context* c, uint64_t* mismatches3, uint32_t* first_bad3) {
    if (!c || !mismatches3) return FYPRT_EINVAL;
    if (c->hostOnly) return c->fail(FYPRT_ESTATE, "fyprt_selftest_math needs a device");
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c));
    struct Scratch { DevBuf<unsigned long long> counts; DevBuf<uint32_t> first; ~Scratch() { counts.release(); first.release(); } } scratch;
    auto& counts = scratch.counts; auto& first = scratch.first;
    HIPCHK(c, counts.alloc(3)); HIPCHK(c, first.alloc(3));
    hipError_t e = hipMemsetAsync(counts.p, 0, 24, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(first.p, 0xFF, 12, c->stream);
    for (int w = 0; w < 3 && e == hipSuccess; ++w) { hipLaunchKernelGGL(k_math_selftest, dim3((uint32_t)c->numCUs * 16u), dim3(256), 0, c->stream, w, counts.p, first.p); e = hipGetLastError(); }
    unsigned long long hc[3] = {0, 0, 0}; uint32_t hf[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(hc, counts.p, 24, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hf, first.p, 12, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    HIPCHK(c, e);
    for (int w = 0; w < 3; ++w) { mismatches3[w] = hc[w]; if (first_bad3) first_bad3[w] = hf[w]; }
    return FYPRT_OK;
}

// MisUtils::ComputeMSE / ComputePSNR (MisUtils.cpp:118-157) of the frame on the device against a host reference image (the benchmark
// workflow of WalnutApp.cpp:826-876 without reading the frame back): RGB channels of the 8-bit images, this context's rows.
int fyprt_compare_image(fyprt_context* c, const uint32_t* reference_rgba8, int flip_reference_rows, double* mse, double* psnr) {
    if (!c || !reference_rgba8 || !mse) return FYPRT_EINVAL;
    if (c->hostOnly || c->W == 0) return c->fail(FYPRT_ESTATE, "fyprt_compare_image before fyprt_resize");
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c));
    const size_t n = (size_t)c->W * c->H;
    if (c->refImage.n != n) HIPCHK(c, c->refImage.alloc(n));
    if (upload(c, c->refImage.p, reference_rgba8, n * 4)) return FYPRT_EHIP;
    HIPCHK(c, hipMemsetAsync(c->rayCounter.p + 31, 0, 8, c->stream));            // (the last, unused counter word serves as the accumulator)
    const uint32_t* img = c->externalImage ? c->externalImage : c->image.p;
    hipLaunchKernelGGL(k_image_sqdiff, dim3((uint32_t)c->numCUs * 4u), dim3(256), 0, c->stream, img, c->refImage.p, c->W, c->H, c->rowBegin, c->rowEnd, flip_reference_rows ? 1 : 0, c->rayCounter.p + 31);
    HIPCHK(c, hipGetLastError());
    unsigned long long tot = 0;
    HIPCHK(c, hipMemcpyAsync(&tot, c->rayCounter.p + 31, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *mse = (double)tot / ((double)((size_t)(c->rowEnd - c->rowBegin) * c->W) * 3.0);
    if (psnr) *psnr = (*mse == 0.0) ? (double)INFINITY : 10.0 * std::log10(255.0 * 255.0 / *mse);
    return FYPRT_OK;
}

int fyprt_image_device_ptr(fyprt_context* c, void** p) { if (!c || !p) return FYPRT_EINVAL; *p = c->externalImage ? (void*)c->externalImage : (void*)c->image.p; return FYPRT_OK; }
int fyprt_set_external_image(fyprt_context* c, void* p) { if (!c) return FYPRT_EINVAL; c->externalImage = (uint32_t*)p; return FYPRT_OK; }
int fyprt_stream(fyprt_context* c, void** s) { if (!c || !s) return FYPRT_EINVAL; *s = (void*)c->stream; return FYPRT_OK; }

int fyprt_read_buffer(fyprt_context* c, int which, void* dst, size_t bytes) {
    if (!c || !dst) return FYPRT_EINVAL;
    if (c->hostOnly) return c->fail(FYPRT_ESTATE, "host-only context has no device buffers");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, sync_all(c));
    const void* src = nullptr; size_t n = 0;
    switch (which) {
        case FYPRT_BUF_ACCUM: src = c->accum.p; n = c->accum.bytes(); break;
        case FYPRT_BUF_IMAGE: src = c->externalImage ? c->externalImage : c->image.p; n = c->image.bytes(); break;
        case FYPRT_BUF_PAYLOAD: src = c->payload.p; n = c->payload.bytes(); break;
        case FYPRT_BUF_DEPTH: src = c->depth.p; n = c->depth.bytes(); break;
        case FYPRT_BUF_NORMAL: src = c->normalFlip ? c->normalB.p : c->normalA.p; n = c->normalA.bytes(); break;   // after the flip "prev" = frame just rendered
        case FYPRT_BUF_DI_RESERVOIR: src = c->di.p; n = c->di.bytes(); break;
        case FYPRT_BUF_DI_PREV: src = c->diPrev.p; n = c->diPrev.bytes(); break;
        case FYPRT_BUF_GI_RESERVOIR: src = c->gi.p; n = c->gi.bytes(); break;
        case FYPRT_BUF_GI_PREV: src = c->giPrev.p; n = c->giPrev.bytes(); break;
        default: return c->fail(FYPRT_EINVAL, "fyprt_read_buffer: unknown buffer");
    }
    if (which == FYPRT_BUF_GI_RESERVOIR || which == FYPRT_BUF_GI_PREV) {
        // the device keeps the 72-byte reservoirs padded to 80 aligned bytes (rt_device.h): hand out the reference layout
        const size_t npx = (size_t)c->W * c->H;
        std::vector<GIRes> rec(npx);
        HIPCHK(c, hipMemcpy(rec.data(), src, npx * sizeof(GIRes), hipMemcpyDeviceToHost));
        const size_t cnt = std::min(npx, bytes / kGIResBytes);
        for (size_t p = 0; p < cnt; ++p) std::memcpy((char*)dst + p * kGIResBytes, &rec[p], kGIResBytes);
        return FYPRT_OK;
    }
    if (c->lastTech == FYPRT_RESTIR_DI && (which == FYPRT_BUF_NORMAL || which == FYPRT_BUF_DI_RESERVOIR || which == FYPRT_BUF_DI_PREV)) {
        // ReSTIR DI keeps normal + reservoir packed in 32-byte records (DIRec); unpack into the reference layouts
        const size_t npx = (size_t)c->W * c->H;
        std::vector<DIRec> rec(npx);
        const DIRec* dsrc = (which == FYPRT_BUF_DI_PREV) ? (c->dprevFlip ? c->dprevB.p : c->dprevA.p) : c->drec.p;   // after the flip "read" = just written
        HIPCHK(c, hipMemcpy(rec.data(), dsrc, npx * sizeof(DIRec), hipMemcpyDeviceToHost));
        if (which == FYPRT_BUF_NORMAL) {
            std::vector<float> out(npx * 2);
            for (size_t p = 0; p < npx; ++p) { out[2 * p] = rec[p].nx; out[2 * p + 1] = rec[p].ny; }
            std::memcpy(dst, out.data(), std::min(bytes, out.size() * 4));
        } else {
            std::vector<DIRes> out(npx);
            for (size_t p = 0; p < npx; ++p) { out[p].index = rec[p].index; out[p].W = rec[p].W; out[p].pdf = rec[p].pdf; out[p].wSum = rec[p].wSum; out[p].M = rec[p].M; }
            std::memcpy(dst, out.data(), std::min(bytes, out.size() * sizeof(DIRes)));
        }
        return FYPRT_OK;
    }
    if (bytes > n) bytes = n;
    if (bytes) HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return FYPRT_OK;
}

int fyprt_reset_frame_index(fyprt_context* c) { if (!c) return FYPRT_EINVAL; c->frameIndex = 1; return FYPRT_OK; }
uint32_t fyprt_frame_index(const fyprt_context* c) { return c ? c->frameIndex : 0; }

int fyprt_export_bvh(fyprt_context* c, void* nodes64, uint32_t* node_count, void* tris48, uint32_t* tri_count, int32_t* root_ref, uint32_t* max_stack) {
    if (!c) return FYPRT_EINVAL;
    if (!c->haveScene) return c->fail(FYPRT_ESTATE, "fyprt_export_bvh before fyprt_upload_scene");
    if (c->hostBvhStale && !c->hostOnly) {              // the device refitted the tree: read it back
        HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c));
        { const int rc = download_nodes(c); if (rc != FYPRT_OK) return rc; }
        if (!c->hostBvh.tris.empty()) HIPCHK(c, hipMemcpy(c->hostBvh.tris.data(), c->leafTris.p, c->hostBvh.tris.size() * 48, hipMemcpyDeviceToHost));
        c->hostBvhStale = false;
    }
    const rth::SceneBVH& b = c->hostBvh;
    if (nodes64) std::memcpy(nodes64, b.nodes.data(), b.nodes.size() * 64);
    if (tris48) std::memcpy(tris48, b.tris.data(), b.tris.size() * 48);
    if (node_count) *node_count = (uint32_t)b.nodes.size();
    if (tri_count) *tri_count = (uint32_t)b.tris.size();
    if (root_ref) *root_ref = b.rootRef;
    if (max_stack) *max_stack = b.levels;
    return FYPRT_OK;
}

int fyprt_export_lighttrees(fyprt_context* c, fyprt_lighttree_node* tlas, uint32_t* tlas_count, uint32_t* tlas_root, fyprt_lighttree_node* blas,
                            uint32_t* blas_total, uint32_t* blas_first, uint32_t* blas_count, uint32_t* blas_root) {
    if (!c) return FYPRT_EINVAL;
    if (!c->haveScene) return c->fail(FYPRT_ESTATE, "fyprt_export_lighttrees before fyprt_upload_scene");
    const rth::LightTrees& l = c->hostLt;
    if (tlas) std::memcpy(tlas, l.tlas.data(), l.tlas.size() * sizeof(fyprt_lighttree_node));
    if (blas) std::memcpy(blas, l.blas.data(), l.blas.size() * sizeof(fyprt_lighttree_node));
    if (tlas_count) *tlas_count = (uint32_t)l.tlas.size();
    if (tlas_root) *tlas_root = l.tlasRoot;
    if (blas_total) *blas_total = (uint32_t)l.blas.size();
    if (blas_first) std::memcpy(blas_first, l.first.data(), l.first.size() * 4);
    if (blas_count) std::memcpy(blas_count, l.count.data(), l.count.size() * 4);
    if (blas_root) std::memcpy(blas_root, l.root.data(), l.root.size() * 4);
    return FYPRT_OK;
}

int fyprt_get_tuning(fyprt_context* c, int key, int* value) {
    if (!c || !value || key < 0 || key >= 24) return FYPRT_EINVAL;
    *value = (key == 8) ? effective_stack_budget(c) : (key == 2 && c->tuning[2] <= 0) ? c->traceOcc : c->tuning[key];   // key 2: residency found at the last DI frame
    return FYPRT_OK;
}

int fyprt_set_tuning(fyprt_context* c, int key, int value) {
    if (!c || key < 0 || key >= 24) return FYPRT_EINVAL;
    // ranges: a value outside them could hang the persistent kernels (refill threshold above the wave size: no lane is ever
    // refilled) or index past a buffer, so it is refused here instead of trusted
    static const int lo[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    static const int hi[24] = {2, 1, 16, 1, 65536, 64, 64, 64, 31, 4096, 65536, 1, 2, 1, 2, 2, 1024, 2, 1, 2, 64, 0, 0, 0};
    if (value < lo[key] || value > hi[key])
        return c->fail(FYPRT_EINVAL, "fyprt_set_tuning: key " + std::to_string(key) + " accepts " + std::to_string(lo[key]) + ".." + std::to_string(hi[key]));
    c->tuning[key] = value;
    return FYPRT_OK;
}

int fyprt_set_ray_counting(fyprt_context* c, int enabled) { if (!c) return FYPRT_EINVAL; c->countRays = enabled != 0; return FYPRT_OK; }

// The two parts of a ReSTIR frame as separate calls, for a host that moves the halo rows between the bands itself (part 1, then its
// own exchange of the buffers fyprt_multi.h lists, then part 2).  Asynchronous.
int fyprt_render_part(fyprt_context* c, const fyprt_settings* s, int part) {
    if (!c || !s || (part != 1 && part != 2)) return FYPRT_EINVAL;
    return enqueue_frame(c, s, true, part);
}

#include "fyprt_multi.h"

}  // extern "C"
