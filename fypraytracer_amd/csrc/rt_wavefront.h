// ReSTIR DI Part 2 as a two-stage wavefront (the north-star's "wavefront ballot / prefix-sum ray compaction and
// persistent-thread work stealing"):
//
//   k_di_part2_setup  one thread per pixel: spatial reuse, light-point sample, BRDF — everything of
//                     PerPixel_ReSTIR_DI_Part2 (R.cu:1875-2007, :2033-2038) except the shadow ray; every pixel of the band
//                     appends one 64-byte task to a queue (wave ballot + prefix popcount, LDS prefix over the 4 waves, ONE
//                     atomic per workgroup): a shadow task, or — for a pixel Part 1 finished (sky / emitter) — a task
//                     without a ray that only carries the pixel's colour to the epilogue.
//   k_di_part2_trace  persistent waves: every lane owns one in-flight shadow ray; when >= refillLanes lanes of a wave are
//                     idle, the wave runs their fused epilogues together (visibility select, accumulate, tonemap, pack)
//                     and refills exactly those lanes from its claimed chunk of the queue, so SIMD lanes stay busy although
//                     shadow rays differ 10x in length (lane utilisation of the one-thread-per-pixel Part 2: 35 %).
//                     Chunks: the first one of every wave is static, the rest are stolen from a shared head and shrink
//                     towards the end of the queue (see the kernel).
//
// All epilogues of a wavefront frame run in the trace kernel, and Part 1 touches neither image nor accumulation
// (DevFrame::p1Mode): that is what lets the host run Part 1 + setup of the next frame on a second stream beside it.
//
// Same arithmetic, same per-ray traversal (trace_shadow of rt_device.h, incl. the closest-hit fallback), so every
// output bit and every instrumentation count is unchanged — only the mapping of rays to lanes differs.
#pragma once
#include "rt_kernels.h"

namespace rt {

struct ShadowQueue {
    float4* tasks; uint32_t* counters; uint32_t chunk, refillLanes, staticChunks, minChunk;
    // light-sorted mode (tuning key 3): tasks are slotted per setup workgroup, a counting sort over kSortBins light bins
    // produces `sorted` (task slots in bin order) without a single global atomic
    uint32_t sortMode, numGroups; uint32_t* counts; uint8_t* keys; uint16_t* hist; uint32_t* binOffset; uint32_t* binTotal; uint32_t* sorted;
};
constexpr uint32_t kNoRayTask = 0xFFFFFFFFu;   // light-triangle field of a task that carries only a finished pixel's colour
constexpr uint32_t kSortBins = 64;   // counters[0] = tail (tasks appended), [1] = head (tasks taken)
constexpr int kRefillLanes = 16;

// ---- spatial reuse with SPECULATIVE neighbour gathers (R.cu:1913-1941).
// The reference's loop is a chain of dependent gathers: neighbour k's pixel comes from the random stream, and the stream advances by
// one more draw when neighbour k-1 was ACCEPTED (its reservoir update draws a number) — so the address of gather k is only known after
// gather k-1 has returned and been tested.  But acceptance depends on geometry alone (depth, normal), and the stream position before
// neighbour k only on HOW MANY of the k earlier neighbours were accepted: 2k + a draws, a = 0..k.  So every address the loop can
// possibly visit is known up front — 1 + 2 + ... + N candidates (15 for N = 5) — and all of them are fetched at once (16-byte hot
// half of the record: depth, normal, light index); the accept / reject walk then runs on registers, and the second halves (W, pdf, M)
// of the <= N accepted records are fetched in one more round.  Two memory round trips instead of N dependent ones; every number
// (addresses, random draws, update order) is the serial loop's.  N > kSpecNeighbors falls back to the serial loop.
// MEASURED SLOWER (tuning key 14, default off): 0.259 vs 0.225 ms on the bench frame — 15 hot quads in flight cost 152 VGPRs (3 waves per
// SIMD instead of 7) and three times the gather requests, while at 7 waves the dependent round trips were already hidden; the kernel
// moves 0.85 GB of HBM in 0.225 ms and is bound by that and by the gather request rate, not by latency (profiles/README.md r02).
constexpr int kSpecNeighbors = 5;
RT_DEV uint32_t neighbor_from_states(const DevCamera& cam, uint32_t W, uint32_t x, uint32_t y, uint32_t radius, uint32_t s1, uint32_t s2) {   // == neighbor_index with explicit draws
    float ox = 2.0f * ((float)s1 * 0x1p-32f) - 1.0f, oy = 2.0f * ((float)s2 * 0x1p-32f) - 1.0f;
    ox = (float)(uint32_t)(x + (uint32_t)(int)(ox * (float)radius));
    oy = (float)(uint32_t)(y + (uint32_t)(int)(oy * (float)radius));
    ox = __builtin_fmaxf(0.0f, __builtin_fminf((float)cam.W - 1.0f, ox));
    oy = __builtin_fmaxf(0.0f, __builtin_fminf((float)cam.H - 1.0f, oy));
    return (uint32_t)ox + (uint32_t)oy * W;
}
// S: the reservoir being built (already holding the pixel's own sample), Z its normalisation count; returns the stream state after the loop
RT_DEV uint32_t spatial_reuse_speculative(const DevCamera& cam, const DevFrame& fr, const DevSettings& st, uint32_t x, uint32_t y, const Payload& pp, uint32_t seed, DIRes& S, uint32_t& Z) {
    const uint32_t N = st.numNeighbors;
    uint32_t xs[3 * kSpecNeighbors + 1];                               // xs[j] = the stream state after j draws
    xs[0] = seed;
#pragma unroll
    for (int j = 1; j <= 3 * kSpecNeighbors; ++j) xs[j] = pcg_hash(xs[j - 1]);
    float4 hot[kSpecNeighbors * (kSpecNeighbors + 1) / 2]; uint32_t nis[kSpecNeighbors * (kSpecNeighbors + 1) / 2];
#pragma unroll
    for (int k = 0; k < kSpecNeighbors; ++k)
#pragma unroll
        for (int a = 0; a <= k; ++a) {
            const int c = k * (k + 1) / 2 + a;
            nis[c] = neighbor_from_states(cam, fr.W, x, y, st.radius, xs[2 * k + a + 1], xs[2 * k + a + 2]);
            hot[c] = ((uint32_t)k < N) ? reinterpret_cast<const float4*>(fr.drec + nis[c])[0] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    // the walk: which candidate each neighbour really is, and whether it passes the geometry test
    uint32_t acc = 0;                                                  // neighbours accepted so far
    uint32_t useNi[kSpecNeighbors]; bool use[kSpecNeighbors]; uint32_t draw[kSpecNeighbors], lightIdx[kSpecNeighbors];
#pragma unroll
    for (int k = 0; k < kSpecNeighbors; ++k) {
        float4 h = hot[k * (k + 1) / 2]; uint32_t ni = nis[k * (k + 1) / 2]; uint32_t d = xs[2 * k + 3];
#pragma unroll
        for (int a = 1; a <= k; ++a) if (acc == (uint32_t)a) { h = hot[k * (k + 1) / 2 + a]; ni = nis[k * (k + 1) / 2 + a]; d = xs[2 * k + a + 3]; }
        const float nd = h.x, pdp = pp.hitDistance;
        f2 nn; nn.x = h.y; nn.y = h.z;
        const bool ok = (uint32_t)k < N && !((nd > 1.1f * pdp || nd < 0.9f * pdp) || (double)dot(nrm3(pp), oct_decode(nn)) < 0.906);
        use[k] = ok; useNi[k] = ni; draw[k] = d; lightIdx[k] = (uint32_t)__float_as_int(h.w);
        acc += ok ? 1u : 0u;
    }
    // second halves of the accepted records, all in flight together
    float4 cold[kSpecNeighbors];
#pragma unroll
    for (int k = 0; k < kSpecNeighbors; ++k) cold[k] = use[k] ? reinterpret_cast<const float4*>(fr.drec + useNi[k])[1] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < kSpecNeighbors; ++k) {
        if (!use[k]) continue;
        const float Wn = cold[k].x, pdf = cold[k].y; const uint32_t M = (uint32_t)__float_as_int(cold[k].w);
        // di_update(S, index, (pdf * W) * M, M, pdf, seed) with the draw the serial loop makes at this point
        const float w = (pdf * Wn) * (float)M;
        S.wSum += w; S.M += M;
        if ((float)draw[k] * 0x1p-32f < w / S.wSum) { S.index = lightIdx[k]; S.pdf = pdf; }
        Z += pdf > 0.0f ? M : 0u;
    }
    uint32_t out = xs[2 * kSpecNeighbors];                             // state after N neighbours with `acc` accepted: xs[2N + acc]
#pragma unroll
    for (int j = 0; j <= 3 * kSpecNeighbors; ++j) if ((uint32_t)j == 2u * N + acc) out = xs[j];
    return out;
}

// MODE 0: one dependent 32-byte gather per neighbour (default).  1: speculative gathers (above).  2: the north-star's "reservoir
// neighbourhood in LDS" — the workgroup's 16x16 tile + `radius` pixels all round (76 x 76 for radius 30) of the 12-byte hot fields
// (depth, octahedral normal) staged in LDS (69 KB), the geometry test served from there, the rest of an ACCEPTED neighbour's record
// fetched from memory.  Measured (tuning key 14 = 2) and NOT the default: see the numbers in profiles/README.md r02 — the tile samples
// 5 x 256 of its 5776 window entries, so staging moves 4.5x the bytes the gathers touch, and 69 KB of LDS leave two workgroups per CU.
constexpr int kWinMax = 16 + 2 * 30;
// (MODE 0 compiles to 74 VGPRs = 6 waves per SIMD; capped to the 72 of 7 waves it spills two registers to scratch — r03, tools/kernel_resources.py -DRT_SETUP_WAVES=...)
#ifndef RT_SETUP_WAVES
#define RT_SETUP_WAVES
#endif
template <int MODE>
__global__ __launch_bounds__(kBlock) RT_SETUP_WAVES void k_di_part2_setup(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st, ShadowQueue q) {
    constexpr bool SPECULATIVE = MODE == 1;
    uint32_t x, y;
    const bool inside = pixel_of_thread(fr, fr.rowBegin, fr.rowEnd, x, y);
    const uint32_t i = x + y * fr.W;
    __shared__ float s_win[MODE == 2 ? 3 * kWinMax * kWinMax : 1];
    __shared__ uint32_t s_org[4];
    uint32_t wx0 = 0, wy0 = 0, ww = 0, wh = 0;
    if (MODE == 2) {
        if (threadIdx.x == 0u) { s_org[0] = inside ? x : 0xFFFFFFFFu; s_org[1] = y; }
        __syncthreads();
        const bool tileOk = s_org[0] != 0xFFFFFFFFu && st.radius <= 30u;      // (thread 0 owns the tile's first pixel: outside => the whole tile is)
        if (tileOk) {
            const uint32_t tx0 = s_org[0], ty0 = s_org[1], R = st.radius;
            wx0 = tx0 > R ? tx0 - R : 0u; wy0 = ty0 > R ? ty0 - R : 0u;
            const uint32_t wx1 = (tx0 + 16u + R < fr.W) ? tx0 + 16u + R : fr.W, wy1 = (ty0 + 16u + R < fr.H) ? ty0 + 16u + R : fr.H;
            ww = wx1 - wx0; wh = wy1 - wy0;
            for (uint32_t k = threadIdx.x; k < ww * wh; k += (uint32_t)kBlock) {
                const uint32_t px = wx0 + k % ww, py = wy0 + k / ww;
                const float4 h = reinterpret_cast<const float4*>(fr.drec + ((size_t)py * fr.W + px))[0];
                s_win[k] = h.x; s_win[kWinMax * kWinMax + k] = h.y; s_win[2 * kWinMax * kWinMax + k] = h.z;
            }
        }
        __syncthreads();
    }
    // Which pixels continue into Part 2 is the reference's sentinel test (Renderer.cu:2787) — here derived from the payload
    // with Part 1's own expressions instead of read back from the image: a pixel that saw the sky or an emitter is finished
    // (R.cu:1650-1668) and becomes a task WITHOUT a ray that only carries its colour to the epilogue, so that Part 1 writes
    // neither image nor accumulation and every epilogue of the frame runs in one place (the trace kernel).
    bool live = false, noRay = false;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), Lvis = splat3(0.0f), Lsky = splat3(0.0f); uint32_t ti = kNoRayTask, lightSlot = 0; float tLight = -1.0f;
    Payload pp; Mat hm;
    if (inside) {
        pp = fr.payload[i];
        if (pp.hitDistance < 0.0f) { noRay = true; Lvis = st.sky; }
        else {
            hm = load_mat(sc, tri_material(sc, pp.objectIndex));
            if (length(emission(hm)) > 0.0f) { noRay = true; Lvis = emission(hm); } else live = true;
        }
    }
    if (live) {
        uint32_t seed = i * (fr.frameIndex + 213u + st.randSeed);
        const DIRec own = load_rec(fr.drec + i);
        DIRes R = rec_reservoir(own);
        const f3 pd = ray_direction(cam, x, y);
        if (st.useSpatial) {
            uint32_t Z = 0; DIRes S = di_empty();
            { const float pdf = R.pdf; di_update(S, R.index, (pdf * R.W) * (float)R.M, R.M, pdf, seed); Z += pdf > 0.0f ? R.M : 0u; }
            if (SPECULATIVE && st.numNeighbors <= (uint32_t)kSpecNeighbors) seed = spatial_reuse_speculative(cam, fr, st, x, y, pp, seed, S, Z);
            else if (MODE == 2 && ww != 0u) for (uint32_t n = 0; n < st.numNeighbors; ++n) {
                const uint32_t ni = neighbor_index(cam, fr.W, x, y, st.radius, seed);
                const uint32_t nxp = ni % fr.W, nyp = ni / fr.W;
                float nd; f2 nn;
                if (nxp - wx0 < ww && nyp - wy0 < wh) { const uint32_t k = (nyp - wy0) * ww + (nxp - wx0); nd = s_win[k]; nn.x = s_win[kWinMax * kWinMax + k]; nn.y = s_win[2 * kWinMax * kWinMax + k]; }
                else { const float4 h = reinterpret_cast<const float4*>(fr.drec + ni)[0]; nd = h.x; nn.x = h.y; nn.y = h.z; }     // the unsigned wrap lands outside the window
                const float pdp = pp.hitDistance;
                if ((nd > 1.1f * pdp || nd < 0.9f * pdp) || (double)dot(nrm3(pp), oct_decode(nn)) < 0.906) continue;
                const DIRes N = rec_reservoir(load_rec(fr.drec + ni));          // accepted: the reservoir half comes from memory
                const float pdf = N.pdf;
                di_update(S, N.index, (pdf * N.W) * (float)N.M, N.M, pdf, seed);
                Z += pdf > 0.0f ? N.M : 0u;
            }
            else for (uint32_t n = 0; n < st.numNeighbors; ++n) {
                const uint32_t ni = neighbor_index(cam, fr.W, x, y, st.radius, seed);
                const DIRec nb = load_rec(fr.drec + ni);                       // one 32-byte gather per neighbour
                const float nd = nb.hitDistance, pdp = pp.hitDistance;
                f2 nn; nn.x = nb.nx; nn.y = nb.ny;
                const DIRes N = rec_reservoir(nb);
                if ((nd > 1.1f * pdp || nd < 0.9f * pdp) || (double)dot(nrm3(pp), oct_decode(nn)) < 0.906) continue;
                const float pdf = N.pdf;
                di_update(S, N.index, (pdf * N.W) * (float)N.M, N.M, pdf, seed);
                Z += pdf > 0.0f ? N.M : 0u;
            }
            const float m = 1.0f / (float)Z;
            S.W = S.pdf > 0.0f ? (1.0f / S.pdf) * (m * S.wSum) : 0.0f;
            R = S;
        }
        lightSlot = R.index;
        const float4* LR = sc.lightRecs + (size_t)R.index * 3;
        const float4 l0 = LR[0], l1 = LR[1], l2 = LR[2];
        ti = (uint32_t)__float_as_int(l1.w);
        const TriGeom g = load_tri(sc, ti);
        const f3 ep = tri_random_point(g, seed);
        f3 dir = ep - pos3(pp);
        const float dist = length(pos3(pp) - ep);
        dir = dir / dist;
        const f3 albedo = sample_albedo(sc, hm, pp.u, pp.v);
        const f3 brdf = eval_brdf(nrm3(pp), -pd, dir, albedo, hm.metallic, hm.roughness);
        const float cx = gmax(dot(dir, nrm3(pp)), 0.0f);
        const float cy = gmax(dot(-dir, mk3(l1.x, l1.y, l1.z)), 0.0f);
        const float sa = l0.w * (dist * dist);
        const f3 T = ((brdf * cx) * cy) / sa;
        const f3 lem = mk3(l2.x, l2.y, l2.z);
        if (length(lem) > 0.0f) { Lvis = T * lem; Lvis = Lvis * R.W; }                         // R.cu:2018-2027
        Lsky = T * st.sky;                                                                     // R.cu:2028-2031
        ro = pos3(pp) + nrm3(pp) * 1e-12f; rd = dir;
        if (st.skipDeadRays && zero3(Lvis) && zero3(Lsky)) ti = kNoRayTask;                    // black whatever the ray finds: a task without a ray (colour Lvis = 0)
        // the shadow ray's distance to its own light triangle (trace_shadow's first step), computed HERE at full lane utilisation so that
        // a refill of the persistent trace kernel is two loads and three reciprocals (<= 0: the ray misses its light -> closest-hit fallback)
        else tLight = light_tri_distance(sc, ti, ro, rd);
        fr.depth[i] = pp.hitDistance;
        { f2 on; on.x = own.nx; on.y = own.ny; store_rec(fr.dprevWrite + i, pp.hitDistance, on, R); }
    }
    // ---- compaction of the live lanes: ballot + prefix popcount inside a wave, a 4-entry LDS prefix across the waves
    __shared__ uint32_t s_count[kBlock / 64];
    __shared__ uint32_t s_base;
    __shared__ uint32_t s_hist[kSortBins];
    const bool queued = live || noRay;
    const unsigned long long mask = __ballot(queued);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0u) s_count[wave] = (uint32_t)__popcll(mask);
    if (q.sortMode && threadIdx.x < kSortBins) s_hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t nLive = s_count[0] + s_count[1] + s_count[2] + s_count[3];
    uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    for (uint32_t k = 0; k < wave; ++k) rank += s_count[k];
    uint32_t slot;
    if (q.sortMode) {
        // slotted storage (256 slots per workgroup, no global atomic) + this workgroup's row of the light histogram
        slot = blockIdx.x * (uint32_t)kBlock + rank;
        const uint32_t key = (uint32_t)(((unsigned long long)lightSlot * kSortBins) / (sc.emissiveCount ? sc.emissiveCount : 1u));
        if (queued) { atomicAdd(&s_hist[key], 1u); q.keys[slot] = (uint8_t)key; }
        __syncthreads();
        if (threadIdx.x < kSortBins) q.hist[(size_t)blockIdx.x * kSortBins + threadIdx.x] = (uint16_t)s_hist[threadIdx.x];
        if (threadIdx.x == 0u) q.counts[blockIdx.x] = nLive;
    } else {
        // compact queue: ONE atomic per workgroup on the tail (a single counter word saturates near 88 atomics/us:
        // one per wave = 34 560 per 1080p frame cost 0.4 ms)
        if (threadIdx.x == 0u) s_base = nLive ? atomicAdd(q.counters + 0, nLive) : 0u;
        __syncthreads();
        slot = s_base + rank;
    }
    if (queued) {
        float4* t = q.tasks + (size_t)slot * 4;
        t[0] = make_float4(ro.x, ro.y, ro.z, tLight);
        t[1] = make_float4(rd.x, rd.y, rd.z, __int_as_float((int)ti));
        t[2] = make_float4(Lvis.x, Lvis.y, Lvis.z, __int_as_float((int)i));      // the pixel travels with both candidate radiances:
        t[3] = make_float4(Lsky.x, Lsky.y, Lsky.z, __int_as_float((int)i));      // the epilogue loads exactly one of them
    }
}

// Column scan of the histogram matrix hist[group][bin]: binOffset[group][bin] = tasks of `bin` in earlier groups,
// binTotal[bin] = tasks of `bin` overall.  One workgroup per bin.
__global__ __launch_bounds__(kBlock) void k_di_sort_scan(ShadowQueue q) {
    __shared__ uint32_t s_part[kBlock];
    const uint32_t bin = blockIdx.x, t = threadIdx.x;
    const uint32_t per = (q.numGroups + kBlock - 1u) / kBlock, g0 = t * per, g1 = (g0 + per < q.numGroups) ? g0 + per : q.numGroups;
    uint32_t sum = 0;
    for (uint32_t g = g0; g < g1; ++g) sum += q.hist[(size_t)g * kSortBins + bin];
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < (uint32_t)kBlock; d <<= 1) {          // Hillis–Steele inclusive scan of the 256 partials
        const uint32_t add = (t >= d) ? s_part[t - d] : 0u;
        __syncthreads();
        s_part[t] += add;
        __syncthreads();
    }
    uint32_t run = s_part[t] - sum;                                 // exclusive prefix of this thread's range
    for (uint32_t g = g0; g < g1; ++g) { q.binOffset[(size_t)g * kSortBins + bin] = run; run += q.hist[(size_t)g * kSortBins + bin]; }
    if (t == (uint32_t)kBlock - 1u) q.binTotal[bin] = s_part[t];
}
// Scatter: every task's final position = start of its bin + tasks of that bin in earlier groups + rank inside the group.
__global__ __launch_bounds__(kBlock) void k_di_sort_scatter(ShadowQueue q) {
    __shared__ uint32_t s_start[kSortBins];
    __shared__ uint32_t s_rank[kSortBins];
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    if (t < kSortBins) s_rank[t] = 0u;
    if (t == 0u) {
        uint32_t run = 0;
        for (uint32_t b = 0; b < kSortBins; ++b) { s_start[b] = run; run += q.binTotal[b]; }
        if (g == 0u) q.counters[0] = run;                           // total number of tasks (queue tail)
    }
    __syncthreads();
    if (t < q.counts[g]) {
        const uint32_t slot = g * (uint32_t)kBlock + t, key = q.keys[slot];
        const uint32_t r = atomicAdd(&s_rank[key], 1u);
        q.sorted[s_start[key] + q.binOffset[(size_t)g * kSortBins + key] + r] = slot;
    }
}

// per-lane in-flight ray of the persistent trace kernel
struct LaneRay {
    f3 o, d; RayPk pk; float tL, cut; uint32_t lightTri, task;      // task: queue slot; pixel index and the two candidate radiances
    int32_t cur; int top; int32_t hitTri; bool closestMode; uint32_t nBox, nTri, nNode;   // are re-read from it at the epilogue (saves 9 VGPRs)
};

RT_DEV void lane_push(int32_t* lds, int& top, int32_t v) { lds[top * kBlock] = v; ++top; }
RT_DEV int32_t lane_pop(int32_t* lds, int& top) { --top; return lds[top * kBlock]; }

// register budget of the persistent trace kernels: 6 waves per SIMD (80 VGPRs); left alone the compiler lands a few registers above that
// boundary and loses a resident workgroup per CU.  (The ReSTIR DI kernel also fits 72 VGPRs = 7 waves without a spill, and the 22 KB LDS
// stack of the bench tree lets a CU hold 7 workgroups — measured slower, 0.458 vs 0.438 ms: profiles/README.md r02.)
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES __attribute__((amdgpu_waves_per_eu(6)))
#endif
template <bool COUNT>
RT_DEV void di_part2_trace_body(const DevScene& sc, const DevFrame& fr, const ShadowQueue& q, int32_t* s_stack) {
    int32_t* lds = s_stack + threadIdx.x;
    const float4* top4 = stage_top_nodes(sc.nodes, sc.topCount, sc.stackBudget, s_stack);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = q.counters[0];
    // Work distribution: every wave owns the first `first` tasks of its own slice statically (wave w: [w * first, (w + 1) * first)),
    // further chunks are stolen from a shared head that starts behind the static slices.  `first` is key 9 x chunk when there is
    // plenty of work and an even share when the queue is shorter than that (a narrow multi-GPU band): then no atomic is
    // issued at all.  Without the static part all 6144 waves of the grid hit the one head word at start-up: 70 us of
    // serialised atomics (88 per us) before the last wave had anything to do — 0.07 of the kernel's 0.53 ms on the full frame,
    // most of its 0.21 ms floor on small bands.
    const uint32_t nWaves = gridDim.x * (uint32_t)(kBlock / 64), myWave = blockIdx.x * (uint32_t)(kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t share = (total + nWaves - 1u) / nWaves;        // an even share of the queue
    const uint32_t chunk = q.chunk;                                // granularity of the dynamic part
    const uint32_t want1 = chunk * q.staticChunks;                 // tasks a wave owns statically when there is plenty (tuning key 9 x chunk)
    const uint32_t first = share < want1 ? (share < 16u ? 16u : share) : want1;
    const uint32_t dynBase = nWaves * first;
    const bool hasDyn = dynBase < total;                           // otherwise the static chunks cover the whole queue: no atomics at all
    LaneRay r; r.cur = kExit; r.top = 0; r.nBox = 0; r.nTri = 0; r.nNode = 0;
    bool active = false;                                           // lane owns a ray that is still being traced
    bool pending = false;                                          // lane's ray is finished, its pixel epilogue not yet run
    uint32_t outcome = 0;                                          // of the finished ray: 0 occluded, 1 light visible, 2 nothing hit (sky)
    bool more = total != 0u;                                       // wave-uniform: tasks may remain (in the queue or in this wave's chunk)
    uint32_t chunkNext = myWave * first < total ? myWave * first : total;     // wave-uniform: this wave's claimed range of the queue
    uint32_t chunkEnd = (myWave + 1u) * first < total ? (myWave + 1u) * first : total;
    while (true) {
        // ---------------- refill: idle lanes take the next tasks of the wave's chunk; a new chunk is stolen from the queue head when it runs dry
        const unsigned long long idle = __ballot(!active);
        if ((more && (uint32_t)__popcll(idle) >= q.refillLanes) || __ballot(active) == 0ull) {
            // finished lanes run the fused epilogue together (accumulate, tonemap, pack) — batched here so that it
            // executes once per >= kRefillLanes rays instead of once per finished ray
            if (pending) {
                const float4 L = q.tasks[(size_t)r.task * 4 + (outcome == 2u ? 3 : 2)];
                const uint32_t pixel = (uint32_t)__float_as_int(L.w);
                const f3 radiance = (outcome != 0u) ? mk3(L.x, L.y, L.z) : splat3(0.0f);
                epilogue(fr, pixel, rgb1(radiance)); pending = false;
            }
        }
        if (more && (uint32_t)__popcll(idle) >= q.refillLanes) {
            if (chunkNext >= chunkEnd && hasDyn) {
                uint32_t base = 0;
                // guided self-scheduling: full chunks while the queue is long, smaller ones towards its end (a 128-task chunk
                // is ~150 us of work for a wave, a third of the kernel); the remaining length is taken from the wave's own
                // previous claim, so no extra access to the head word is needed
                uint32_t size = (total - chunkEnd) / nWaves;
                size = size < q.minChunk ? q.minChunk : (size > chunk ? chunk : size);
                if (lane == 0u) base = dynBase + atomicAdd(q.counters + 1, size);
                base = (uint32_t)__shfl((int)base, 0);
                chunkNext = base < total ? base : total;
                chunkEnd = (base + size < total) ? base + size : total;
            }
            const uint32_t slot = chunkNext + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t want = (uint32_t)__popcll(idle), avail = chunkEnd - chunkNext;
            chunkNext += (want < avail) ? want : avail;
            more = chunkNext < chunkEnd || (hasDyn && chunkEnd < total);
            if (!active && slot < chunkEnd) {
                r.task = q.sortMode ? q.sorted[slot] : slot;
                const float4* t = q.tasks + (size_t)r.task * 4;
                const float4 t0 = t[0], t1 = t[1];
                r.o = mk3(t0.x, t0.y, t0.z);
                r.d = mk3(t1.x, t1.y, t1.z); r.lightTri = (uint32_t)__float_as_int(t1.w);
                if (r.lightTri == kNoRayTask) { pending = true; outcome = 1u; }          // a finished pixel: straight to the epilogue with its colour (t[2])
                else {
                r.pk = make_raypk(r.o, safe_inv(r.d.x), safe_inv(r.d.y), safe_inv(r.d.z));
                const float tL = t0.w;                                  // distance to the light triangle, from the setup kernel
                r.closestMode = !(tL > 0.0f);
                r.tL = r.closestMode ? 3.402823466e+38f : tL;
                r.cut = r.tL * 1.000001f;
                r.hitTri = r.closestMode ? -1 : (int32_t)r.lightTri;
                if (COUNT) { r.nBox = 0; r.nTri = 1; r.nNode = 0; }
                r.top = 0; lane_push(lds, r.top, kExit);
                r.cur = (sc.triCount == 0 || ray_not_finite(r.o, r.d)) ? kExit : sc.rootRef;   // (a non-finite ray also fails the light test above: closest mode, miss)
                active = true;
                }
            }
        }
        if (__ballot(active) == 0ull) { if (!more && __ballot(pending) == 0ull) break; else continue; }   // (tasks without a ray leave lanes pending but not active)
        // ---------------- traverse until enough lanes have finished to make a refill worthwhile
        while (true) {
            // inner nodes
            bool walk = active && r.cur >= 0;                  // (one compare per round serves the loop condition and the quorum ballot)
            const uint32_t quorum = quorum_of(sc.nodeQuorum, (uint32_t)__popcll(__ballot(active)));
            while (walk) {
                { Stack st; st.lds = lds; st.top = r.top; st.top4 = top4; st.topCount = sc.topCount; r.cur = node_step<COUNT>(sc.nodes, sc.stackBudget, r.cur, r.pk, r.cut, st, r.nBox, r.nNode); r.top = st.top; }
                walk = r.cur >= 0;
                if ((uint32_t)__popcll(__ballot(walk)) < quorum) break;
            }
            // leaves
            if (active && r.cur < 0 && r.cur != kExit) {
                const uint32_t code = (uint32_t)~r.cur, first = code >> 2, cnt = (code & 3u) + 1u;
                bool occluded = false;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const float4* tp = sc.leafTris + (size_t)(first + k) * 3;
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    const uint32_t id = (uint32_t)__float_as_int(c.y);
                    if (!r.closestMode && id == r.lightTri) continue;
                    if (COUNT) r.nTri += 1;
                    const f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
                    const f3 hh = cross(r.d, e2);
                    const float det = dot(e1, hh), f = 1.0f / det;
                    const f3 s = r.o - v0;
                    const float u = f * dot(s, hh);
                    if (u < 0.0f || u > 1.0f) continue;
                    const f3 qq = cross(s, e1);
                    const float v = f * dot(r.d, qq);
                    if (v < 0.0f || (u + v) > 1.0f) continue;
                    const float t = f * dot(e2, qq);
                    if (t > 0.0001f && t < r.tL) {
                        r.hitTri = (int32_t)id;
                        if (r.closestMode) { r.tL = t; r.cut = t * 1.000001f; }
                        else { occluded = true; break; }
                    }
                }
                r.cur = occluded ? kExit : lane_pop(lds, r.top);
            }
            // finished lanes: select the pixel's radiance, park it until the next batched epilogue
            if (active && r.cur == kExit) {
                outcome = 0u;
                if (r.hitTri == (int32_t)r.lightTri) outcome = 1u;                      // R.cu:2016-2027: Lvis
                else if (r.closestMode && r.hitTri < 0) outcome = 2u;                   // R.cu:2028-2031: Lsky
                pending = true;
                if (COUNT) {
                    // same totals as trace_shadow: the fallback counts its light test, then a full closest-hit ray
                    atomicAdd(sc.rayCounter + 0, 1ull); atomicAdd(sc.rayCounter + 1, (unsigned long long)r.nBox);
                    atomicAdd(sc.rayCounter + 2, (unsigned long long)r.nTri);
                    atomicAdd(sc.rayCounter + 3, (unsigned long long)((r.closestMode && r.hitTri < 0) ? 0 : 1));
                    atomicAdd(sc.rayCounter + 4, (unsigned long long)r.nNode);
                }
                active = false;
            }
            const unsigned long long act = __ballot(active);
            if (act == 0ull) break;
            if (more && (64u - (uint32_t)__popcll(act)) >= q.refillLanes) break;
        }
    }
}

template <bool COUNT> __global__ void k_di_part2_trace(DevScene sc, DevFrame fr, ShadowQueue q);
template <> __global__ __launch_bounds__(kBlock) RT_TRACE_WAVES void k_di_part2_trace<false>(DevScene sc, DevFrame fr, ShadowQueue q) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch
    di_part2_trace_body<false>(sc, fr, q, s_stack);
}
template <> __global__ __launch_bounds__(kBlock) void k_di_part2_trace<true>(DevScene sc, DevFrame fr, ShadowQueue q) {   // instrumented: no register cap
    extern __shared__ int32_t s_stack[];
    di_part2_trace_body<true>(sc, fr, q, s_stack);
}

// =====================================================================================================================
// Generic ray queue of the wavefront path engine (rt_paths.h): every technique but ReSTIR DI runs as
//   primary kernel -> [ shade step -> k_trace_rays ] x N -> last shade step,
// where a shade step (one thread per live path, no traversal state) consumes the results of the rays it emitted in the previous
// step and emits the next ones, compacted with ballot + prefix popcount.  A ray record is three quads:
//   q0 = origin.xyz | owner pixel      q1 = direction.xyz | mode      q2 = mode arguments
//   mode = kRayClosest : closest hit                                   -> result (t, u, v, triangle | -1)
//          kRayVisible : ReSTIR GI visibility query, q2 = (dist, tol)  -> result.x = 1 iff a hit lies in [dist - tol, dist + tol] and none before
//          otherwise   : shadow ray towards light triangle `mode`; q2.x = distance to the light triangle along the ray, computed by the
//                        emitting shade step at full lane utilisation (<= 0: the ray misses its own light -> exact closest-hit fallback)
//                                                                       -> result (hitDistance | -1, -, -, objectIndex) == trace_shadow()
// k_trace_rays: persistent waves, every lane owns one in-flight ray, idle lanes are refilled from the wave's claimed chunk of the
// queue (same guided self-scheduling as the ReSTIR DI trace kernel); it holds nothing but traversal state: a refill is two loads
// and three reciprocals, a finished ray is one 16-byte store.
constexpr uint32_t kRayClosest = 0xFFFFFFFEu, kRayVisible = 0xFFFFFFFDu;
// kRayNone: a slot that holds no ray — NEE's shadow ray when the prepared direct term is exactly zero, so that nothing the ray could find would
// change the pixel (DevSettings::skipDeadRays; 45 % of the NEE shadow rays of the bench scene: lights the surface faces away from).  Answered
// at refill as "light not reached" (-1, -, -, -1) without being traced or counted.
constexpr uint32_t kRayNone = 0xFFFFFFFCu;
struct TraceQueue {
    const float4* rays; float4* hits; const uint32_t* count; uint32_t raysPer; uint32_t* head;
    uint32_t chunk, refillLanes, staticChunks, minChunk;
};

template <bool COUNT>
RT_DEV void trace_rays_body(const DevScene& sc, const TraceQueue& q, int32_t* s_stack) {
    int32_t* lds = s_stack + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = *q.count * q.raysPer;
    const uint32_t nWaves = gridDim.x * (uint32_t)(kBlock / 64), myWave = blockIdx.x * (uint32_t)(kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t share = (total + nWaves - 1u) / nWaves;
    const uint32_t chunk = q.chunk, want1 = chunk * q.staticChunks;
    const uint32_t first = share < want1 ? (share < 16u ? 16u : share) : want1;
    const uint32_t dynBase = nWaves * first;
    const bool hasDyn = dynBase < total;
    f3 o = splat3(0.0f), d = splat3(0.0f); RayPk pk = make_raypk(o, 1.0f, 1.0f, 1.0f);
    float tL = 0.0f, cut = 0.0f, hu = 0.0f, hv = 0.0f; uint32_t mode = kRayClosest, task = 0; int32_t cur = kExit, hitTri = -1; int top = 0;
    bool closestMode = true;
    uint32_t nBox = 0, nTri = 0, nNode = 0;
    bool active = false;
    bool more = total != 0u;
    uint32_t chunkNext = myWave * first < total ? myWave * first : total;
    uint32_t chunkEnd = (myWave + 1u) * first < total ? (myWave + 1u) * first : total;
    while (true) {
        const unsigned long long idle = __ballot(!active);
        if (more && (uint32_t)__popcll(idle) >= q.refillLanes) {
            if (chunkNext >= chunkEnd && hasDyn) {
                uint32_t base = 0;
                uint32_t size = (total - chunkEnd) / nWaves;
                size = size < q.minChunk ? q.minChunk : (size > chunk ? chunk : size);
                if (lane == 0u) base = dynBase + atomicAdd(q.head, size);
                base = (uint32_t)__shfl((int)base, 0);
                chunkNext = base < total ? base : total;
                chunkEnd = (base + size < total) ? base + size : total;
            }
            const uint32_t slot = chunkNext + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t want = (uint32_t)__popcll(idle), avail = chunkEnd - chunkNext;
            chunkNext += (want < avail) ? want : avail;
            more = chunkNext < chunkEnd || (hasDyn && chunkEnd < total);
            if (!active && slot < chunkEnd) {
                task = slot;
                const float4* t = q.rays + (size_t)task * 3;
                const float4 t0 = t[0], t1 = t[1], t2 = t[2];
                o = mk3(t0.x, t0.y, t0.z); d = mk3(t1.x, t1.y, t1.z); mode = (uint32_t)__float_as_int(t1.w);
                pk = make_raypk(o, safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
                hitTri = -1; hu = 0.0f; hv = 0.0f; closestMode = true; tL = 3.402823466e+38f;
                if (mode == kRayVisible) { closestMode = false; hu = t2.x - t2.y; tL = t2.x + t2.y; }     // hu = dist - tol, interval end = dist + tol; hv: 0 nothing yet, 1 found, -1 blocked
                else if (mode != kRayClosest && t2.x > 0.0f) { closestMode = false; tL = t2.x; hitTri = (int32_t)mode; }
                cut = tL * 1.000001f;
                if (COUNT) { nBox = 0; nNode = 0; nTri = (mode != kRayClosest && mode != kRayVisible) ? 1u : 0u; }
                top = 0; lane_push(lds, top, kExit);
                // (a slot without a ray — kRayNone — is a closest-hit record that never starts: it leaves through the ordinary exit with the
                // "nothing hit" answer (-1, 0, 0, -1), uncounted)
                cur = (sc.triCount == 0 || mode == kRayNone || ray_not_finite(o, d)) ? kExit : sc.rootRef;
                active = true;
            }
        }
        if (__ballot(active) == 0ull) { if (!more) break; else continue; }
        while (true) {
            bool walk = active && cur >= 0;
            const uint32_t quorum = quorum_of(sc.nodeQuorum, (uint32_t)__popcll(__ballot(active)));
            while (walk) {
                { Stack st; st.lds = lds; st.top = top; cur = node_step<COUNT>(sc.nodes, sc.stackBudget, cur, pk, cut, st, nBox, nNode); top = st.top; }
                walk = cur >= 0;
                if ((uint32_t)__popcll(__ballot(walk)) < quorum) break;
            }
            if (active && cur < 0 && cur != kExit) {
                const uint32_t code = (uint32_t)~cur, firstTri = code >> 2, cnt = (code & 3u) + 1u;
                bool done = false;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const float4* tp = sc.leafTris + (size_t)(firstTri + k) * 3;
                    if (!closestMode && (uint32_t)__float_as_int(tp[2].y) == mode) continue;     // a shadow ray does not test its own light again
                    float t, u, v; uint32_t id;
                    if (COUNT) nTri += 1;
                    if (!tri_test(tp, o, d, t, u, v, id)) continue;
                    if (mode == kRayVisible) {
                        if (t < hu) { hv = -1.0f; done = true; break; }
                        if (t <= tL) hv = 1.0f;
                    } else if (t < tL) {
                        hitTri = (int32_t)id; tL = t;
                        if (closestMode) { cut = t * 1.000001f; hu = u; hv = v; }
                        else { done = true; break; }
                    }
                }
                cur = done ? kExit : lane_pop(lds, top);
            }
            if (active && cur == kExit) {
                float4 res;
                if (mode == kRayVisible) res = make_float4(hv > 0.0f ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f);
                else res = make_float4(hitTri < 0 ? -1.0f : tL, hu, hv, __int_as_float(hitTri));
                q.hits[task] = res;
                if (COUNT && mode != kRayNone) {
                    atomicAdd(sc.rayCounter + 0, 1ull); atomicAdd(sc.rayCounter + 1, (unsigned long long)nBox);
                    atomicAdd(sc.rayCounter + 2, (unsigned long long)nTri);
                    atomicAdd(sc.rayCounter + 3, (unsigned long long)((mode == kRayVisible) ? (hv > 0.0f ? 1 : 0) : (hitTri < 0 ? 0 : 1)));
                    atomicAdd(sc.rayCounter + 4, (unsigned long long)nNode);
                }
                active = false;
            }
            const unsigned long long act = __ballot(active);
            if (act == 0ull) break;
            if (more && (64u - (uint32_t)__popcll(act)) >= q.refillLanes) break;
        }
    }
}

// The same ray records traced ONE THREAD PER RAY, no persistence: for scenes whose rays are cheap (a few node visits) the refill
// machinery of the persistent kernel costs more than the divergence it removes (banana stand-in, 3 k triangles, 3 node visits per
// ray: 130 us per launch persistent, see profiles/README.md r02).  Chosen by the host from the tree size (tuning key 15).
template <bool COUNT>
RT_DEV float4 trace_one(const DevScene& sc, f3 o, f3 d, uint32_t mode, float a0, float a1, int32_t* ldsBase) {
    float tL = 3.402823466e+38f, hu = 0.0f, hv = 0.0f; int32_t hitTri = -1; bool closestMode = true;
    if (mode == kRayVisible) { closestMode = false; hu = a0 - a1; tL = a0 + a1; }
    else if (mode != kRayClosest && a0 > 0.0f) { closestMode = false; tL = a0; hitTri = (int32_t)mode; }
    float cut = tL * 1.000001f;
    uint32_t nBox = 0, nNode = 0, nTri = (mode != kRayClosest && mode != kRayVisible) ? 1u : 0u;
    if (!(sc.triCount == 0 || ray_not_finite(o, d))) {
        const RayPk pk = make_raypk(o, safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
        Stack st; st.lds = ldsBase; st.top = 0; st.push(kExit);
        int32_t cur = sc.rootRef;
        bool done = false;
        while (!done) {
            bool walk = cur >= 0;
            const uint32_t quorum = quorum_of(sc.nodeQuorum, (uint32_t)__popcll(__ballot(true)));
            while (walk) {
                cur = node_step<COUNT>(sc.nodes, sc.stackBudget, cur, pk, cut, st, nBox, nNode);
                walk = cur >= 0;
                if ((uint32_t)__popcll(__ballot(walk)) < quorum) break;
            }
            if (cur >= 0) continue;
            if (cur == kExit) break;
            const uint32_t code = (uint32_t)~cur, firstTri = code >> 2, cnt = (code & 3u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const float4* tp = sc.leafTris + (size_t)(firstTri + k) * 3;
                if (!closestMode && (uint32_t)__float_as_int(tp[2].y) == mode) continue;
                float t, u, v; uint32_t id;
                if (COUNT) nTri += 1;
                if (!tri_test(tp, o, d, t, u, v, id)) continue;
                if (mode == kRayVisible) {
                    if (t < hu) { hv = -1.0f; done = true; break; }
                    if (t <= tL) hv = 1.0f;
                } else if (t < tL) {
                    hitTri = (int32_t)id; tL = t;
                    if (closestMode) { cut = t * 1.000001f; hu = u; hv = v; }
                    else { done = true; break; }
                }
            }
            cur = st.pop();
        }
    }
    if (COUNT) {
        atomicAdd(sc.rayCounter + 0, 1ull); atomicAdd(sc.rayCounter + 1, (unsigned long long)nBox); atomicAdd(sc.rayCounter + 2, (unsigned long long)nTri);
        atomicAdd(sc.rayCounter + 3, (unsigned long long)((mode == kRayVisible) ? (hv > 0.0f ? 1 : 0) : (hitTri < 0 ? 0 : 1)));
        atomicAdd(sc.rayCounter + 4, (unsigned long long)nNode);
    }
    if (mode == kRayVisible) return make_float4(hv > 0.0f ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f);
    return make_float4(hitTri < 0 ? -1.0f : tL, hu, hv, __int_as_float(hitTri));
}
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_rays_simple(DevScene sc, TraceQueue q) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch
    const uint32_t total = *q.count * q.raysPer;
    for (uint32_t base = blockIdx.x * (uint32_t)kBlock; base < total; base += gridDim.x * (uint32_t)kBlock) {   // (block-uniform trips: node_step's ballots see whole waves)
        const uint32_t j = base + threadIdx.x;
        if (j < total) {
            const float4* t = q.rays + (size_t)j * 3;
            const float4 t0 = t[0], t1 = t[1], t2 = t[2];
            if ((uint32_t)__float_as_int(t1.w) == kRayNone) q.hits[j] = make_float4(-1.0f, 0.0f, 0.0f, __int_as_float(-1));
            else q.hits[j] = trace_one<COUNT>(sc, mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), (uint32_t)__float_as_int(t1.w), t2.x, t2.y, s_stack + threadIdx.x);
        }
    }
}

template <bool COUNT> __global__ void k_trace_rays(DevScene sc, TraceQueue q);
template <> __global__ __launch_bounds__(kBlock) RT_TRACE_WAVES void k_trace_rays<false>(DevScene sc, TraceQueue q) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch
    trace_rays_body<false>(sc, q, s_stack);
}
template <> __global__ __launch_bounds__(kBlock) void k_trace_rays<true>(DevScene sc, TraceQueue q) {                       // instrumented: no register cap
    extern __shared__ int32_t s_stack[];
    trace_rays_body<true>(sc, q, s_stack);
}

}  // namespace rt
