// Multi-GPU layer of the C ABI (include/fyprt.h, "multi-GPU" section) — included at the end of fyprt.hip, inside extern "C".
//
// The frame is split into row bands, one context (one GPU) per band.  Two ways to get the ReSTIR
// records of the `radius` rows either side of a band ("halo") that Part 2's spatial reuse reads:
//   recompute (halo mode 0)  every band runs Part 1 on its halo rows too (fyprt_set_rows(..., halo)); no data crosses the fabric
//                            before the image gather; halo rows have no temporal history, so from frame 2 on pixels near a band
//                            border differ (unbiased) from a single-GPU frame;
//   exchange  (halo mode 1)  every band runs Part 1 on its own rows only; between Part 1 and Part 2 it receives the Part-1 records of
//                            its halo rows from the bands that own them, and before Part 1 their temporal history — 32 B per pixel
//                            and exchange for ReSTIR DI (1.8 MB per neighbour at 1080p), 96 B + 72 B for ReSTIR GI.  Every record is
//                            then computed once, by its owner, with its full history: a static-camera sequence is bit-identical to the
//                            single-GPU sequence on EVERY frame (tests/test_gpu_group.py), and Part 1 does 22 % less work at 8 bands.
//                            The history rows fetched are the spatial halo but at least two (temporal reuse without spatial reuse
//                            still reprojects onto the row above now and then), and a switch between the two ReSTIRs brings every
//                            band's "previous normals" in step BEFORE they travel (tests/test_gpu_group_sequences.py: 400 random
//                            sequences of techniques, settings, moved borders, restarts and interleave switches, all exact).
// Two transports under the same plan (`halo_plan`):
//   fyprt_group_*   one process, one context per GPU (what a C++ host such as the reference's application is): hipMemcpyPeerAsync
//                   between the contexts' buffers, ordered with events — no collective library at all;
//   fyprt_comm_*    one process per GPU (torch.distributed.run, MPI, ...): RCCL — grouped ncclSend / ncclRecv for the halos, grouped
//                   ncclBroadcast (one per band, in place in the full-size image) for the gather.  librccl.so.1 is opened on first use.
#include <dlfcn.h>

extern "C++" {
namespace {

struct RowSpan { uint32_t r0, r1; };
struct HaloXfer { int receiver, owner; uint32_t r0, r1; };

// every (receiver, owner, rows) transfer of one halo exchange, in the canonical order both ends of a pair derive independently
std::vector<HaloXfer> halo_plan(const std::vector<uint32_t>& bounds, uint32_t halo, uint32_t H, bool wrapRow) {
    std::vector<HaloXfer> plan;
    const int n = (int)bounds.size() - 1;
    for (int r = 0; r < n; ++r) {
        const uint32_t b = bounds[r], e = bounds[r + 1];
        if (b >= e) continue;
        const uint32_t lo = b > halo ? b - halo : 0u, hi = (e + halo < H) ? e + halo : H;
        std::vector<RowSpan> need;
        if (lo < b) need.push_back({lo, b});
        if (e < hi) need.push_back({e, hi});
        // the reference's unsigned neighbour wrap (R.cu:1916-1917): an offset above row 0 clamps to the LAST row
        if (wrapRow && b < halo && hi < H) need.push_back({H - 1u, H});
        for (const RowSpan& sp : need)
            for (int j = 0; j < n; ++j) {
                if (j == r) continue;
                const uint32_t a0 = std::max(sp.r0, bounds[j]), a1 = std::min(sp.r1, bounds[j + 1]);
                if (a0 < a1) plan.push_back({r, j, a0, a1});
            }
    }
    return plan;
}

struct XBuf { void* p; size_t bytesPerPixel; };
// what crosses between Part 1 and Part 2 (kind 0) / before Part 1 (kind 1: temporal history), for the technique
std::vector<XBuf> exchange_buffers(fyprt_context* c, int tech, int kind) {
    std::vector<XBuf> v;
    if (tech == FYPRT_RESTIR_DI) {
        if (kind == 0) v.push_back({c->drec.p, sizeof(DIRec)});
        else v.push_back({c->dprevFlip ? (void*)c->dprevB.p : (void*)c->dprevA.p, sizeof(DIRec)});
    } else {
        if (kind == 0) {
            v.push_back({c->gi.p, sizeof(GIRes)}); v.push_back({c->giHot.p, 4 * sizeof(float4)});        // Part 2 reads a neighbour's 64-byte record; the reservoir only of the finally selected sample
            v.push_back({c->normalFlip ? (void*)c->normalA.p : (void*)c->normalB.p, sizeof(f2)});      // this frame's normals: next frame's history normals
        } else {
            v.push_back({c->giPrev.p, sizeof(GIRes)});
            v.push_back({c->normalFlip ? (void*)c->normalB.p : (void*)c->normalA.p, sizeof(f2)});      // the previous normals the history is tested against
        }
    }
    return v;
}
bool is_restir(const fyprt_settings* s) { return s->technique == FYPRT_RESTIR_DI || s->technique == FYPRT_RESTIR_GI; }
// stripes of part `part` of an interleaved split, as (first row, end row) — the same walk as stripe_row_count
template <class F> void for_each_stripe(uint32_t H, uint32_t stripe, uint32_t parts, uint32_t part, F&& f) {
    for (uint64_t r0 = (uint64_t)part * stripe; r0 < H; r0 += (uint64_t)parts * stripe) f((uint32_t)r0, (uint32_t)std::min<uint64_t>(r0 + stripe, H));
}
// rows around a band whose Part-1 records Part 2 reads (spatial reuse: the kernels' uint8 cast of the radius, R.cu:1897), and rows whose
// temporal history Part 1 may be reprojected onto: the spatial halo, at least kHistoryRows (a static camera reprojects a pixel onto
// itself or, by fp32 rounding of an exactly integral coordinate, onto its upper / left neighbour: Camera.cpp:143)
constexpr uint32_t kHistoryRows = 2;
uint32_t spatial_halo(const fyprt_settings* s, int n) { return (is_restir(s) && s->use_spatial_reuse && n > 1) ? ((uint32_t)s->spatial_neighbor_radius & 0xFFu) : 0u; }
uint32_t history_halo(const fyprt_settings* s, int n) { return (is_restir(s) && s->use_temporal_reuse && n > 1) ? std::max(spatial_halo(s, n), kHistoryRows) : 0u; }
void extend_history_rows(fyprt_context* c, int tech, uint32_t halo) {
    uint32_t* h = tech == FYPRT_RESTIR_DI ? c->histDI : c->histGI;
    h[0] = c->rowBegin > halo ? c->rowBegin - halo : 0u; h[1] = (c->rowEnd + halo < c->H) ? c->rowEnd + halo : c->H;
}

// ---- RCCL, opened lazily: a host that never calls fyprt_comm_* does not need the library
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr; int (*CommInitRank)(void**, int, ncclUniqueIdBytes, int) = nullptr; int (*CommDestroy)(void*) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr; int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr; const char* (*GetErrorString)(int) = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = std::string("cannot open librccl.so.1: ") + dlerror(); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId"); CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy"); Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
        Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv"); GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd"); GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!(GetUniqueId && CommInitRank && CommDestroy && Broadcast && Send && Recv && GroupStart && GroupEnd && GetErrorString)) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
} g_rccl;
constexpr int kNcclChar = 0;      // ncclInt8 / ncclChar: transfers are counted in bytes

// The point-to-point operations ONE rank issues inside one ncclGroupStart / ncclGroupEnd section, in issue order.  RCCL matches the
// sends and receives of a pair of ranks in the order each side issues them, so what has to hold — and what tests/test_multigpu_plan.py
// checks on the CPU for every pair, since more than one rank per device cannot run on the one-GPU box — is that rank a's sends to b
// and rank b's receives from a list the same byte counts in the same order.  Both functions are pure: (plan, rank) -> list.
struct CommOp { int recv, peer, buf; size_t offset, bytes; };
std::vector<CommOp> comm_exchange_ops(int rank, const std::vector<HaloXfer>& plan, const std::vector<size_t>& bytesPerPixel, uint32_t W) {
    std::vector<CommOp> ops;
    for (const HaloXfer& x : plan)
        for (size_t b = 0; b < bytesPerPixel.size(); ++b) {
            const size_t off = (size_t)x.r0 * W * bytesPerPixel[b], bytes = (size_t)(x.r1 - x.r0) * W * bytesPerPixel[b];
            if (x.receiver == rank) ops.push_back({1, x.owner, (int)b, off, bytes});
            else if (x.owner == rank) ops.push_back({0, x.receiver, (int)b, off, bytes});
        }
    return ops;
}
// rows that change owner when the band table `oldB` is replaced by `newB`: [r0, r1) owned by j so far, by k from now on
std::vector<CommOp> comm_set_rows_ops(int rank, const std::vector<uint32_t>& oldB, const std::vector<uint32_t>& newB, const std::vector<size_t>& bytesPerPixel, uint32_t W) {
    std::vector<CommOp> ops;
    const int n = (int)oldB.size() - 1;
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j) {
            if (j == k) continue;
            const uint32_t r0 = std::max(newB[k], oldB[j]), r1 = std::min(newB[k + 1], oldB[j + 1]);
            if (r0 >= r1 || (rank != k && rank != j)) continue;
            for (size_t b = 0; b < bytesPerPixel.size(); ++b) {
                const size_t off = (size_t)r0 * W * bytesPerPixel[b], bytes = (size_t)(r1 - r0) * W * bytesPerPixel[b];
                ops.push_back({rank == k ? 1 : 0, rank == k ? j : k, (int)b, off, bytes});
            }
        }
    return ops;
}

}  // namespace
}  // extern "C++"

struct fyprt_group {
    std::vector<fyprt_context*> ctx; std::vector<uint32_t> bounds; int haloMode = 0;
    uint32_t stripeRows = 0, lastStripeRows = 0; bool lastStriped = false;      // interleaved split for the per-pixel techniques; whether (and with which stripes) the last frame used it
    std::vector<hipEvent_t> evP1, evPulled, evFrame, evSync; hipEvent_t evGather = nullptr; bool gatherPending = false; std::string err;
};

int fyprt_group_synchronize(fyprt_group* g);
#define NCCLCHK(c, call) do { const int _r = (call); if (_r != 0) return (c)->fail(FYPRT_EHIP, std::string(#call) + ": " + g_rccl.GetErrorString(_r)); } while (0)
// One group section: every operation of `ops` on the context's stream.  An error inside the section still closes it (an open group
// would swallow every later fyprt_comm_* call of this thread and the next synchronize would wait for ever).
static int comm_issue(fyprt_context* c, const std::vector<CommOp>& ops, const std::vector<void*>& base) {
    NCCLCHK(c, g_rccl.GroupStart());
    int bad = 0;
    for (const CommOp& o : ops) {
        char* p = (char*)base[(size_t)o.buf] + o.offset;
        const int e = o.recv ? g_rccl.Recv(p, o.bytes, kNcclChar, o.peer, c->comm, c->stream) : g_rccl.Send(p, o.bytes, kNcclChar, o.peer, c->comm, c->stream);
        if (e != 0) { bad = e; break; }
    }
    const int endRc = g_rccl.GroupEnd();
    if (bad != 0) return c->fail(FYPRT_EHIP, std::string("ncclSend / ncclRecv: ") + g_rccl.GetErrorString(bad));
    NCCLCHK(c, endRc);
    return FYPRT_OK;
}

// ------------------------------------------------------------------------------------------------ one process, one context per GPU
int fyprt_group_create(fyprt_context** ctxs, int n, const uint32_t* row_bounds, fyprt_group** out) {
    if (!ctxs || n <= 0 || !row_bounds || !out) return FYPRT_EINVAL;
    *out = nullptr;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->hostOnly || ctxs[i]->H == 0) return FYPRT_EINVAL;
        if (ctxs[i]->W != ctxs[0]->W || ctxs[i]->H != ctxs[0]->H) return ctxs[i]->fail(FYPRT_EINVAL, "fyprt_group_create: contexts differ in size");
        if (row_bounds[i] >= row_bounds[i + 1]) return ctxs[i]->fail(FYPRT_EINVAL, "fyprt_group_create: empty or unordered band");
    }
    if (row_bounds[0] != 0 || row_bounds[n] != ctxs[0]->H) return ctxs[0]->fail(FYPRT_EINVAL, "fyprt_group_create: the bands must partition rows 0..height");
    auto* g = new fyprt_group();
    g->ctx.assign(ctxs, ctxs + n); g->bounds.assign(row_bounds, row_bounds + n + 1);
    g->evP1.resize(n); g->evPulled.resize(n); g->evFrame.resize(n); g->evSync.resize(n);
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(ctxs[i]->device);
        (void)hipEventCreateWithFlags(&g->evP1[i], hipEventDisableTiming); (void)hipEventCreateWithFlags(&g->evPulled[i], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&g->evFrame[i], hipEventDisableTiming); (void)hipEventCreateWithFlags(&g->evSync[i], hipEventDisableTiming);
    }
    (void)hipSetDevice(ctxs[0]->device);
    (void)hipEventCreateWithFlags(&g->evGather, hipEventDisableTiming);
    *out = g;
    return FYPRT_OK;
}
void fyprt_group_destroy(fyprt_group* g) {
    if (!g) return;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->ctx[i]->device); (void)sync_all(g->ctx[i]);
        (void)hipEventDestroy(g->evP1[i]); (void)hipEventDestroy(g->evPulled[i]); (void)hipEventDestroy(g->evFrame[i]); (void)hipEventDestroy(g->evSync[i]);
        g->ctx[i]->haloExchange = false;
    }
    if (g->evGather) (void)hipEventDestroy(g->evGather);
    delete g;
}
// New band boundaries (e.g. from fyprt_balance_rows).  Rows that change owner take their accumulation and their temporal history
// (ReSTIR DI records, ReSTIR GI reservoirs + normals) with them — peer copies from the old owner, between two frames — so a sequence
// with moving borders stays what it was.  Synchronises the group.
int fyprt_group_set_rows(fyprt_group* g, const uint32_t* row_bounds) {
    if (!g || !row_bounds) return FYPRT_EINVAL;
    const int n = (int)g->ctx.size();
    const uint32_t W = g->ctx[0]->W;
    if (row_bounds[0] != 0 || row_bounds[n] != g->ctx[0]->H) return g->ctx[0]->fail(FYPRT_EINVAL, "fyprt_group_set_rows: the bands must partition rows 0..height");
    for (int i = 0; i < n; ++i) if (row_bounds[i] >= row_bounds[i + 1]) return g->ctx[i]->fail(FYPRT_EINVAL, "fyprt_group_set_rows: empty or unordered band");
    { const int rc = fyprt_group_synchronize(g); if (rc != FYPRT_OK) return rc; }
    for (int k = 0; k < n; ++k) {
        fyprt_context* c = g->ctx[k];
        HIPCHK(c, hipSetDevice(c->device));
        for (int j = 0; j < n; ++j) {
            if (j == k) continue;
            const uint32_t r0 = std::max(row_bounds[k], g->bounds[j]), r1 = std::min(row_bounds[k + 1], g->bounds[j + 1]);
            if (r0 >= r1) continue;                                   // rows [r0, r1): owned by j so far, by k from now on
            fyprt_context* o = g->ctx[j];
            auto move = [&](void* dst, const void* src, size_t bpp) { return hipMemcpyPeerAsync((char*)dst + (size_t)r0 * W * bpp, c->device, (const char*)src + (size_t)r0 * W * bpp, o->device, (size_t)(r1 - r0) * W * bpp, c->stream); };
            // (the accumulation of an interleaved frame lives in stripes, not in bands: nothing to move — every context keeps its stripes)
            if (!g->lastStriped) HIPCHK(c, move(c->accum.p, o->accum.p, sizeof(float4)));
            HIPCHK(c, move(c->dprevFlip ? c->dprevB.p : c->dprevA.p, o->dprevFlip ? o->dprevB.p : o->dprevA.p, sizeof(DIRec)));
            HIPCHK(c, move(c->giPrev.p, o->giPrev.p, sizeof(GIRes)));
            HIPCHK(c, move(c->normalFlip ? c->normalB.p : c->normalA.p, o->normalFlip ? o->normalB.p : o->normalA.p, sizeof(f2)));
        }
    }
    { const int rc = fyprt_group_synchronize(g); if (rc != FYPRT_OK) return rc; }
    g->bounds.assign(row_bounds, row_bounds + n + 1);
    for (int k = 0; k < n; ++k) {
        fyprt_context* c = g->ctx[k];
        c->histDI[0] = c->histGI[0] = row_bounds[k]; c->histDI[1] = c->histGI[1] = row_bounds[k + 1];
    }
    return FYPRT_OK;
}
int fyprt_group_set_halo_mode(fyprt_group* g, int mode) { if (!g || mode < 0 || mode > 1) return FYPRT_EINVAL; g->haloMode = mode; return FYPRT_OK; }
// Interleaved bands for the techniques without spatial reuse: context i renders the stripes of `stripe_rows` rows with
// index i modulo n — see fyprt_set_row_stripes.  ReSTIR frames keep the contiguous bands.  0 = contiguous bands for everything.
// A change takes effect with the next frame and, like a technique change in the reference (Renderer::ResetFrameIndex), wants the
// accumulation restarted: the rows a context has accumulated are the rows it rendered.
int fyprt_group_set_interleave(fyprt_group* g, uint32_t stripe_rows) {
    if (!g) return FYPRT_EINVAL;
    if (stripe_rows != 0 && (uint64_t)stripe_rows * g->ctx.size() > g->ctx[0]->H) return g->ctx[0]->fail(FYPRT_EINVAL, "fyprt_group_set_interleave: fewer stripes than contexts");
    g->stripeRows = stripe_rows;
    return FYPRT_OK;
}

// One frame on every band (asynchronous: returns when everything is enqueued; fyprt_group_synchronize / fyprt_synchronize wait).
int fyprt_group_render(fyprt_group* g, const fyprt_settings* s) {
    if (!g || !s) return FYPRT_EINVAL;
    const int n = (int)g->ctx.size();
    const uint32_t H = g->ctx[0]->H, W = g->ctx[0]->W;
    const uint32_t halo = spatial_halo(s, n), hhalo = history_halo(s, n);
    const bool exchange = g->haloMode == 1 && (halo > 0 || hhalo > 0);
    const bool striped = g->stripeRows != 0 && n > 1 && !is_restir(s);
    g->lastStriped = striped; g->lastStripeRows = g->stripeRows;
    for (int i = 0; i < n; ++i) {
        fyprt_context* c = g->ctx[i];
        c->rowBegin = g->bounds[i]; c->rowEnd = g->bounds[i + 1]; c->halo = halo; c->rowsSet = true; c->haloExchange = exchange;
        if (striped) { const int rc = fyprt_set_row_stripes(c, g->stripeRows, (uint32_t)n, (uint32_t)i); if (rc != FYPRT_OK) return rc; }
        else c->stripeRows = 0;
        // the previous frame's gather still reads this band's image rows on the root's stream: the new frame's epilogues wait for it
        if (g->gatherPending) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipStreamWaitEvent(c->stream, g->evGather, 0)); }
    }
    g->gatherPending = false;
    if (!exchange) {
        for (int i = 0; i < n; ++i) {
            fyprt_context* c = g->ctx[i];
            const int rc = enqueue_frame(c, s, true);
            if (rc != FYPRT_OK) return rc;
            HIPCHK(c, hipEventRecord(g->evFrame[i], c->stream));
        }
        return FYPRT_OK;
    }
    const std::vector<HaloXfer> plan = halo_plan(g->bounds, halo, H, true), hplan = halo_plan(g->bounds, hhalo, H, false);
    auto pull = [&](int i, const std::vector<HaloXfer>& pl, int kind, const std::vector<hipEvent_t>& ready) -> int {
        fyprt_context* c = g->ctx[i];
        HIPCHK(c, hipSetDevice(c->device));
        const std::vector<XBuf> mine = exchange_buffers(c, s->technique, kind);
        for (const HaloXfer& x : pl) {
            if (x.receiver != i) continue;
            fyprt_context* o = g->ctx[x.owner];
            HIPCHK(c, hipStreamWaitEvent(c->stream, ready[x.owner], 0));
            if (kind == 1) HIPCHK(c, hipStreamWaitEvent(c->stream, g->evSync[x.owner], 0));      // the owner's history is in step with the technique
            const std::vector<XBuf> theirs = exchange_buffers(o, s->technique, kind);
            for (size_t b = 0; b < mine.size(); ++b) {
                const size_t off = (size_t)x.r0 * W * mine[b].bytesPerPixel, bytes = (size_t)(x.r1 - x.r0) * W * mine[b].bytesPerPixel;
                HIPCHK(c, hipMemcpyPeerAsync((char*)mine[b].p + off, c->device, (const char*)theirs[b].p + off, o->device, bytes, c->stream));
            }
        }
        return FYPRT_OK;
    };
    // 0. a switch between the two ReSTIRs: every band brings its "previous normals" in step before anybody fetches them
    for (int i = 0; i < n; ++i) {
        fyprt_context* c = g->ctx[i];
        HIPCHK(c, hipSetDevice(c->device));
        { const int rc = sync_restir_normals(c, s->technique, c->stream); if (rc != FYPRT_OK) return rc; }
        HIPCHK(c, hipEventRecord(g->evSync[i], c->stream));
    }
    // 1. temporal history of the halo rows, then Part 1 on the band's own rows
    for (int i = 0; i < n; ++i) {
        fyprt_context* c = g->ctx[i];
        HIPCHK(c, hipSetDevice(c->device));
        for (int j = 0; j < n; ++j) if (j != i) HIPCHK(c, hipStreamWaitEvent(c->stream, g->evPulled[j], 0));    // last frame's pulls FROM this band are done
        if (s->use_temporal_reuse) { const int rc = pull(i, hplan, 1, g->evFrame); if (rc != FYPRT_OK) return rc; }
        extend_history_rows(c, s->technique, hhalo);
        const int rc = enqueue_frame(c, s, true, 1);
        if (rc != FYPRT_OK) return rc;
        HIPCHK(c, hipEventRecord(g->evP1[i], c->stream));
    }
    // 2. the neighbours' Part-1 records of the halo rows, then Part 2
    for (int i = 0; i < n; ++i) {
        fyprt_context* c = g->ctx[i];
        { const int rc = pull(i, plan, 0, g->evP1); if (rc != FYPRT_OK) return rc; }
        HIPCHK(c, hipEventRecord(g->evPulled[i], c->stream));
    }
    for (int i = 0; i < n; ++i) {
        fyprt_context* c = g->ctx[i];
        HIPCHK(c, hipSetDevice(c->device));
        const int rc = enqueue_frame(c, s, true, 2);
        if (rc != FYPRT_OK) return rc;
        HIPCHK(c, hipEventRecord(g->evFrame[i], c->stream));
    }
    return FYPRT_OK;
}
// The bands' RGBA8 rows into the image of context `root` (the one that presents), on root's stream, after every band's frame.
int fyprt_group_gather(fyprt_group* g, int root) {
    if (!g || root < 0 || root >= (int)g->ctx.size()) return FYPRT_EINVAL;
    fyprt_context* r = g->ctx[root];
    HIPCHK(r, hipSetDevice(r->device));
    uint32_t* dst = r->externalImage ? r->externalImage : r->image.p;
    for (int i = 0; i < (int)g->ctx.size(); ++i) {
        if (i == root) continue;
        fyprt_context* c = g->ctx[i];
        HIPCHK(r, hipStreamWaitEvent(r->stream, g->evFrame[i], 0));
        const uint32_t* src = c->externalImage ? c->externalImage : c->image.p;
        hipError_t e = hipSuccess;
        auto rows = [&](uint32_t r0, uint32_t r1) {
            const size_t off = (size_t)r0 * r->W, cnt = (size_t)(r1 - r0) * r->W;
            if (e == hipSuccess) e = hipMemcpyPeerAsync(dst + off, r->device, src + off, c->device, cnt * 4, r->stream);
        };
        if (g->lastStriped) for_each_stripe(r->H, g->lastStripeRows, (uint32_t)g->ctx.size(), (uint32_t)i, rows);
        else rows(g->bounds[i], g->bounds[i + 1]);
        HIPCHK(r, e);
    }
    HIPCHK(r, hipEventRecord(g->evGather, r->stream));
    g->gatherPending = true;
    return FYPRT_OK;
}
int fyprt_group_synchronize(fyprt_group* g) {
    if (!g) return FYPRT_EINVAL;
    for (fyprt_context* c : g->ctx) { HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c)); }
    return FYPRT_OK;
}

// Cost-balanced bands: new boundaries from the time each band took last frame, assuming a band's cost is spread evenly over its rows
// (piecewise-constant cost density), moved at most `max_shift` rows per boundary and frame, every band at least `min_rows` high.
int fyprt_balance_rows(const uint32_t* row_bounds, const float* band_ms, int n, uint32_t min_rows, uint32_t max_shift, uint32_t* new_bounds) {
    if (!row_bounds || !band_ms || !new_bounds || n <= 0) return FYPRT_EINVAL;
    const uint32_t H = row_bounds[n];
    double total = 0.0;
    for (int i = 0; i < n; ++i) { if (!(band_ms[i] > 0.0f) || row_bounds[i] >= row_bounds[i + 1]) return FYPRT_EINVAL; total += band_ms[i]; }
    new_bounds[0] = 0; new_bounds[n] = H;
    int band = 0; double before = 0.0;                       // cost of the bands entirely above the cursor
    for (int k = 1; k < n; ++k) {
        const double target = total * k / n;
        while (band < n - 1 && before + band_ms[band] < target) { before += band_ms[band]; ++band; }
        const double frac = (target - before) / band_ms[band];
        double row = row_bounds[band] + frac * (row_bounds[band + 1] - row_bounds[band]);
        const double lo = (double)row_bounds[k] - max_shift, hi = (double)row_bounds[k] + max_shift;
        row = std::min(std::max(row, lo), hi);
        uint32_t r = (uint32_t)std::max(0.0, row + 0.5);
        const uint32_t floor_ = new_bounds[k - 1] + min_rows, ceil_ = H - (uint32_t)(n - k) * min_rows;
        if (r < floor_) r = floor_;
        if (r > ceil_) r = ceil_;
        new_bounds[k] = r;
    }
    for (int k = 0; k < n; ++k) if (new_bounds[k] >= new_bounds[k + 1]) return FYPRT_EINVAL;      // min_rows x n exceeds the height
    return FYPRT_OK;
}

// Time of the last frame on this context (sum of its launches' hipEvent durations; synchronises the context).
int fyprt_last_frame_ms(fyprt_context* c, float* ms) {
    if (!c || !ms) return FYPRT_EINVAL;
    float part[4]; uint32_t n = 0;
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, sync_all(c));
    const int rc = fyprt_frame_timings(c, 0, part, &n);
    if (rc != FYPRT_OK) return rc;
    *ms = part[0] + part[1] + part[2] + part[3];
    return FYPRT_OK;
}

// ------------------------------------------------------------------------------------------------ one process per GPU: RCCL
int fyprt_comm_unique_id(void* id128) {
    if (!id128) return FYPRT_EINVAL;
    if (!g_rccl.load()) { g_createError = g_rccl.err; return FYPRT_EHIP; }
    const int r = g_rccl.GetUniqueId(id128);
    if (r != 0) { g_createError = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return FYPRT_EHIP; }
    return FYPRT_OK;
}
int fyprt_comm_init_rank(fyprt_context* c, int world, int rank, const void* id128, const uint32_t* row_bounds) {
    if (!c || !id128 || !row_bounds || world <= 0 || rank < 0 || rank >= world) return FYPRT_EINVAL;
    if (c->hostOnly || c->H == 0) return c->fail(FYPRT_ESTATE, "fyprt_comm_init_rank: resize the context first");
    if (row_bounds[0] != 0 || row_bounds[world] != c->H) return c->fail(FYPRT_EINVAL, "fyprt_comm_init_rank: the bands must partition rows 0..height");
    for (int i = 0; i < world; ++i) if (row_bounds[i] >= row_bounds[i + 1]) return c->fail(FYPRT_EINVAL, "fyprt_comm_init_rank: empty or unordered band");
    if (!g_rccl.load()) return c->fail(FYPRT_EHIP, g_rccl.err);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    ncclUniqueIdBytes id; std::memcpy(id.b, id128, 128);
    NCCLCHK(c, g_rccl.CommInitRank(&c->comm, world, id, rank));
    c->world = world; c->rank = rank; c->bounds.assign(row_bounds, row_bounds + world + 1);
    c->rowBegin = row_bounds[rank]; c->rowEnd = row_bounds[rank + 1]; c->rowsSet = true;
    return FYPRT_OK;
}
// New band boundaries for every rank (collective: every rank calls it with the same table, between two frames).  Rows that change
// owner take their accumulation and temporal history along (grouped ncclSend / ncclRecv), as fyprt_group_set_rows does with peer copies.
int fyprt_comm_set_rows(fyprt_context* c, const uint32_t* row_bounds) {
    if (!c || !row_bounds || !c->comm) return FYPRT_EINVAL;
    if (row_bounds[0] != 0 || row_bounds[c->world] != c->H) return c->fail(FYPRT_EINVAL, "fyprt_comm_set_rows: the bands must partition rows 0..height");
    for (int i = 0; i < c->world; ++i) if (row_bounds[i] >= row_bounds[i + 1]) return c->fail(FYPRT_EINVAL, "fyprt_comm_set_rows: empty or unordered band");
    HIPCHK(c, hipSetDevice(c->device));
    // Between two frames for real: a pipelined ReSTIR DI frame still has Part 1 + setup of the NEXT frame in flight on the front stream,
    // and they read and write the very history rows that move here.  Everything of this context drains first, and the call returns only
    // once the rows have arrived — exactly what fyprt_group_set_rows does around its peer copies.
    HIPCHK(c, sync_all(c));
    std::vector<void*> base; std::vector<size_t> bpp;
    if (!c->commLastStriped) { base.push_back(c->accum.p); bpp.push_back(sizeof(float4)); }     // an interleaved frame's accumulation lives in stripes: nothing to move
    base.push_back(c->dprevFlip ? (void*)c->dprevB.p : (void*)c->dprevA.p); bpp.push_back(sizeof(DIRec));
    base.push_back(c->giPrev.p); bpp.push_back(sizeof(GIRes));
    base.push_back(c->normalFlip ? (void*)c->normalB.p : (void*)c->normalA.p); bpp.push_back(sizeof(f2));
    { const int rc = comm_issue(c, comm_set_rows_ops(c->rank, c->bounds, std::vector<uint32_t>(row_bounds, row_bounds + c->world + 1), bpp, c->W), base); if (rc != FYPRT_OK) return rc; }
    HIPCHK(c, sync_all(c));
    c->bounds.assign(row_bounds, row_bounds + c->world + 1);
    c->rowBegin = c->bounds[c->rank]; c->rowEnd = c->bounds[c->rank + 1];
    c->histDI[0] = c->histGI[0] = c->rowBegin; c->histDI[1] = c->histGI[1] = c->rowEnd;
    return FYPRT_OK;
}
int fyprt_comm_set_halo_mode(fyprt_context* c, int mode) { if (!c || mode < 0 || mode > 1) return FYPRT_EINVAL; c->commHaloMode = mode; return FYPRT_OK; }
// fyprt_group_set_interleave for one process per GPU (every rank calls it with the same value)
int fyprt_comm_set_interleave(fyprt_context* c, uint32_t stripe_rows) {
    if (!c || !c->comm) return FYPRT_EINVAL;
    if (stripe_rows != 0 && (uint64_t)stripe_rows * c->world > c->H) return c->fail(FYPRT_EINVAL, "fyprt_comm_set_interleave: fewer stripes than ranks");
    c->commStripeRows = stripe_rows;
    return FYPRT_OK;
}
void fyprt_comm_destroy(fyprt_context* c) { if (c && c->comm && g_rccl.lib) { (void)hipSetDevice(c->device); (void)sync_all(c); (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; c->haloExchange = false; } }

static int comm_exchange(fyprt_context* c, int tech, const std::vector<HaloXfer>& plan, int kind) {
    const std::vector<XBuf> bufs = exchange_buffers(c, tech, kind);
    std::vector<void*> base; std::vector<size_t> bpp;
    for (const XBuf& b : bufs) { base.push_back(b.p); bpp.push_back(b.bytesPerPixel); }
    return comm_issue(c, comm_exchange_ops(c->rank, plan, bpp, c->W), base);
}
// One frame of this rank's band (asynchronous).  Every rank of the communicator must call it with the same settings.
int fyprt_comm_render(fyprt_context* c, const fyprt_settings* s) {
    if (!c || !s) return FYPRT_EINVAL;
    if (!c->comm) return c->fail(FYPRT_ESTATE, "fyprt_comm_render before fyprt_comm_init_rank");
    const uint32_t halo = spatial_halo(s, c->world), hhalo = history_halo(s, c->world);
    const bool exchange = c->commHaloMode == 1 && (halo > 0 || hhalo > 0);
    c->rowBegin = c->bounds[c->rank]; c->rowEnd = c->bounds[c->rank + 1]; c->halo = halo; c->rowsSet = true; c->haloExchange = exchange;
    c->commLastStriped = c->commStripeRows != 0 && c->world > 1 && !is_restir(s); c->commLastStripeRows = c->commStripeRows;
    if (c->commLastStriped) { const int rc = fyprt_set_row_stripes(c, c->commStripeRows, (uint32_t)c->world, (uint32_t)c->rank); if (rc != FYPRT_OK) return rc; }
    else c->stripeRows = 0;
    if (!exchange) return enqueue_frame(c, s, true);
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc = sync_restir_normals(c, s->technique, c->stream); if (rc != FYPRT_OK) return rc; }      // before the history travels
    if (s->use_temporal_reuse) { const int rc = comm_exchange(c, s->technique, halo_plan(c->bounds, hhalo, c->H, false), 1); if (rc != FYPRT_OK) return rc; }
    extend_history_rows(c, s->technique, hhalo);
    { const int rc = enqueue_frame(c, s, true, 1); if (rc != FYPRT_OK) return rc; }
    { const int rc = comm_exchange(c, s->technique, halo_plan(c->bounds, halo, c->H, true), 0); if (rc != FYPRT_OK) return rc; }
    return enqueue_frame(c, s, true, 2);
}
// The image gather of the north-star design: every band's RGBA8 rows, in place in the full-size image (the context's own or the
// external one), one grouped ncclBroadcast per band on the context's stream.  root < 0: every rank ends up with the whole frame;
// root >= 0: only that rank does (ncclSend / ncclRecv).
int fyprt_comm_gather(fyprt_context* c, int root) {
    if (!c) return FYPRT_EINVAL;
    if (!c->comm) return c->fail(FYPRT_ESTATE, "fyprt_comm_gather before fyprt_comm_init_rank");
    if (root >= c->world) return c->fail(FYPRT_EINVAL, "fyprt_comm_gather: root out of range");
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t* img = c->externalImage ? c->externalImage : c->image.p;
    NCCLCHK(c, g_rccl.GroupStart());
    int bad = 0;
    for (int r = 0; r < c->world; ++r) {
        auto rows = [&](uint32_t r0, uint32_t r1) {
            uint32_t* band = img + (size_t)r0 * c->W;
            const size_t bytes = (size_t)(r1 - r0) * c->W * 4;
            int e = 0;
            if (root < 0) e = g_rccl.Broadcast(band, band, bytes, kNcclChar, r, c->comm, c->stream);
            else if (r != root) {
                if (c->rank == r) e = g_rccl.Send(band, bytes, kNcclChar, root, c->comm, c->stream);
                else if (c->rank == root) e = g_rccl.Recv(band, bytes, kNcclChar, r, c->comm, c->stream);
            }
            if (e != 0 && bad == 0) bad = e;
        };
        if (c->commLastStriped) for_each_stripe(c->H, c->commLastStripeRows, (uint32_t)c->world, (uint32_t)r, rows);      // one transfer per stripe, all in one group
        else rows(c->bounds[r], c->bounds[r + 1]);
    }
    const int endRc = g_rccl.GroupEnd();
    if (bad != 0) return c->fail(FYPRT_EHIP, std::string("fyprt_comm_gather: ") + g_rccl.GetErrorString(bad));
    NCCLCHK(c, endRc);
    return FYPRT_OK;
}

// Test hook: the transfer plan of a halo exchange (receiver, owner, first row, end row per entry); returns the number of entries.
int fyprt_halo_plan(const uint32_t* row_bounds, int n, uint32_t halo, uint32_t height, int wrap_row, uint32_t* out4, int capacity) {
    if (!row_bounds || n <= 0) return -1;
    const std::vector<HaloXfer> plan = halo_plan(std::vector<uint32_t>(row_bounds, row_bounds + n + 1), halo, height, wrap_row != 0);
    for (int k = 0; k < (int)plan.size() && k < capacity && out4; ++k) {
        out4[4 * k] = (uint32_t)plan[k].receiver; out4[4 * k + 1] = (uint32_t)plan[k].owner; out4[4 * k + 2] = plan[k].r0; out4[4 * k + 3] = plan[k].r1;
    }
    return (int)plan.size();
}

// Test hook: the point-to-point operations rank `rank` issues in one group section — kind 0: a halo exchange over `bounds` (halo rows,
// wrap row as fyprt_halo_plan); kind 1: fyprt_comm_set_rows from `bounds` to `new_bounds`.  out5 = (is_recv, peer, buffer, offset, bytes)
// per operation; returns the number of operations.  Pure host arithmetic: no device, no RCCL.
int fyprt_comm_ops(int kind, const uint32_t* bounds, const uint32_t* new_bounds, int n, uint32_t halo, uint32_t height, int wrap_row, uint32_t width,
                   int rank, const uint32_t* bytes_per_pixel, int nbuf, uint64_t* out5, int capacity) {
    if (!bounds || n <= 0 || rank < 0 || rank >= n || !bytes_per_pixel || nbuf <= 0 || (kind == 1 && !new_bounds)) return -1;
    const std::vector<uint32_t> b(bounds, bounds + n + 1);
    const std::vector<size_t> bpp(bytes_per_pixel, bytes_per_pixel + nbuf);
    const std::vector<CommOp> ops = kind == 0 ? comm_exchange_ops(rank, halo_plan(b, halo, height, wrap_row != 0), bpp, width)
                                               : comm_set_rows_ops(rank, b, std::vector<uint32_t>(new_bounds, new_bounds + n + 1), bpp, width);
    for (int k = 0; k < (int)ops.size() && k < capacity && out5; ++k) {
        out5[5 * k] = (uint64_t)ops[k].recv; out5[5 * k + 1] = (uint64_t)ops[k].peer; out5[5 * k + 2] = (uint64_t)ops[k].buf; out5[5 * k + 3] = ops[k].offset; out5[5 * k + 4] = ops[k].bytes;
    }
    return (int)ops.size();
}
