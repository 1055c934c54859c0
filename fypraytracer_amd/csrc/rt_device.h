// Device-side scene / frame descriptors and the traversal core (gfx950).
//
// HBM layout — everything a ray touches is a 16-byte-aligned quad stream:
//   nodes     : 4 x float4 per node   (both child boxes + child refs, 64 B)
//   leafTris  : 3 x float4 per record (v0, e1, e2, original triangle index, 48 B), leaf order
//   triPos    : 3 x float4 per ORIGINAL triangle (p0|material, p1, p2)           — light sampling
//   triShade  : 4 x float4 per ORIGINAL triangle (n0|uv0.x, n1|uv0.y, n2|uv1.x, uv1.y uv2.x uv2.y material)
//   materials : 3 x float4 per material (albedo|mapinfo, roughness metallic power -, emission colour)
// so a hit costs one dependent gather (triShade) instead of the reference's
// triangle -> 3 vertex indices -> 3 x 32-byte vertices chain (Renderer.cu:2399-2413).
#pragma once
#include "rt_math.h"

namespace rt {

struct DevTexture { const uint32_t* pixels; uint32_t width, height, _pad; };

struct DevLTNode {   // == fyprt_lighttree_node (80 B)
    float energy; uint32_t numEmitters, left, rightOrEmitter, isLeaf; float axis[3], theta_o, theta_e; float lo[3], hi[3], centroid[3]; uint32_t _pad;
};

struct DevScene {
    const float4* nodes; const float4* leafTris; int32_t rootRef; uint32_t triCount;
    const float4* triPos; const float4* triShade; const float4* mats;
    const DevTexture* textures; uint32_t textureCount;
    const uint32_t* emissive; uint32_t emissiveCount; const float4* lightRecs;
    const DevLTNode* ltTlas; uint32_t ltTlasCount, ltTlasRoot;
    const DevLTNode* ltBlas; const uint32_t* ltFirst; const uint32_t* ltCount; const uint32_t* ltRoot; const uint32_t* ltLeafOfTri;
    unsigned long long* rayCounter;   // instrumented kernel variants (template COUNT) only: [0] rays [1] box tests [2] triangle tests [3] hits [4] node visits
    uint32_t topCount;                // nodes [0, topCount) are also kept in LDS by the traversal kernels (Stack::top4)
    int32_t stackBudget;              // pending-entry budget of node_step's stack rule (kStackBudget; tests lower it to exercise resume entries)
    uint32_t nodeQuorum;              // leave the inner-node loop when fewer lanes than this are still in it (0 = never)
};

struct DevCamera { m4 invProj, invView, prevProjView; f3 position; uint32_t W, H; };

struct Payload { float hitDistance; float px, py, pz; float nx, ny, nz; float u, v; int32_t objectIndex; };   // Ray.h:13-22 (40 B)
struct DIRes { uint32_t index; float W, pdf, wSum; uint32_t M; };                                             // 20 B
struct GISample { float vp[3]; float vn[2]; float sp[3]; float sn[2]; float Lo[3]; uint32_t seed; float pdf; };
// The reference's ReSTIR_GI_Reservoir is 72 bytes (8-byte aligned: a record is nine dwordx2 loads and straddles cache lines).  On the
// device it is padded to 80 bytes and aligned to 16: five dwordx4 loads per record — Part 2 gathers two of them per neighbour step.
// fyprt_read_buffer hands out the reference's 72-byte layout.
struct alignas(16) GIRes { GISample s; float W; uint32_t M; float wSum; float pad[2]; };
constexpr size_t kGIResBytes = 72;
static_assert(sizeof(Payload) == 40 && sizeof(DIRes) == 20 && sizeof(GIRes) == 80 && sizeof(GISample) == 60, "layout");

// ReSTIR DI per-pixel record: everything a *neighbour* (spatial reuse) or the *next frame* (temporal reuse) reads
// about a pixel, packed into one aligned 32-byte line half — primary hit distance, octahedral normal, reservoir —
// so a gather costs one 32-byte access instead of three scattered ones into the reference's separate 40-byte
// payload / 8-byte normal / 20-byte reservoir arrays (R.cu:1924-1934, :1764-1765).  Unpacked again by fyprt_read_buffer.
struct DIRec { float hitDistance, nx, ny; uint32_t index; float W, pdf, wSum; uint32_t M; };
static_assert(sizeof(DIRec) == 32, "layout");

struct DevFrame {
    float4* accum; uint32_t* image; Payload* payload; float* depth; f2* normalPrev; f2* normalCur;
    DIRes* di; DIRes* diPrev; GIRes* gi; GIRes* giPrev;
    float4* giHot;   // ReSTIR GI: what Part 2 reads about a NEIGHBOUR, one aligned 64-byte record (4 x float4) per pixel, written by Part 1:
                     //   [0] primary hit distance, octahedral normal, |Lo| of the Part-1 reservoir — the 16 bytes the acceptance test needs
                     //   [1] visible point, M   [2] sample point, weightSum   [3] octahedral sample normal — what a merge needs
                     // so a rejected neighbour costs one 16-byte gather and an accepted one the rest of the same cache line, instead of
                     // gathers into the payload (40 B), normal (8 B) and reservoir (80 B, two or three lines) arrays
    DIRec* drec; const DIRec* dprevRead; DIRec* dprevWrite;   // DI: this frame's records; previous frame's (read) / next frame's history (write)
    uint32_t W, H, frameIndex, rowBegin, rowEnd, tileOrder;
    uint32_t histBegin, histEnd;   // rows whose ReSTIR history (previous frame) this context holds: the band it rendered last frame
    uint32_t stripeRows, stripeParts, stripePart;   // interleaved multi-GPU split of the per-pixel techniques (fyprt_set_row_stripes): this context
                       // renders the stripes of `stripeRows` rows whose index is `stripePart` modulo `stripeParts`; stripeRows 0 = the contiguous band
    uint32_t p1Mode;   // ReSTIR DI Part 1: 0 = writes the image sentinel and the epilogue of finished pixels itself (reference protocol, R.cu:2746-2750);
                       // 1 = touches neither image nor accumulation: the wavefront Part 2 derives "finished" from the payload and runs every epilogue
};

struct DevSettings {   // RenderingSettings.h:5-22 with the kernel-side uint8 casts already applied
    f3 sky; uint32_t maxBounces, sampleCount, candidateCount, randSeed, useTemporal, useSpatial, historyLimit, numNeighbors, radius;
    uint32_t skipDeadRays;   // ReSTIR DI Part 2: a shadow ray whose pixel is black in EVERY outcome (both candidate radiances exactly zero: reservoir weight 0,
                             // a light facing away, ...) is not traced — same pixel, fewer rays (tuning key 18; the reference traces it, R.cu:2010-2031)
};
RT_DEV bool zero3(f3 v) { return v.x == 0.0f && v.y == 0.0f && v.z == 0.0f; }      // +-0 only: a NaN or an infinity is not zero

struct Hit { float t, u, v; int32_t tri; };

RT_DEV DIRec load_rec(const DIRec* p) {
    const float4* q = reinterpret_cast<const float4*>(p); const float4 a = q[0], b = q[1];
    DIRec r; r.hitDistance = a.x; r.nx = a.y; r.ny = a.z; r.index = (uint32_t)__float_as_int(a.w);
    r.W = b.x; r.pdf = b.y; r.wSum = b.z; r.M = (uint32_t)__float_as_int(b.w); return r;
}
RT_DEV void store_rec(DIRec* p, float hitDistance, f2 n, const DIRes& r) {
    float4* q = reinterpret_cast<float4*>(p);
    q[0] = make_float4(hitDistance, n.x, n.y, __int_as_float((int)r.index));
    q[1] = make_float4(r.W, r.pdf, r.wSum, __int_as_float((int)r.M));
}
RT_DEV DIRes rec_reservoir(const DIRec& c) { DIRes r; r.index = c.index; r.W = c.W; r.pdf = c.pdf; r.wSum = c.wSum; r.M = c.M; return r; }

RT_DEV f3 pos3(const Payload& p) { return mk3(p.px, p.py, p.pz); }
RT_DEV f3 nrm3(const Payload& p) { return mk3(p.nx, p.ny, p.nz); }

// ---------------------------------------------------------------------------------------------
// Traversal stack: entirely in LDS, laid out [entry][thread] so that a wave's push/pop is one conflict-free
// ds_write_b32 / ds_read_b32 (32 KB per 256-thread workgroup, 5 workgroups per CU — the kernels are VGPR-limited to
// <= 5 waves/SIMD anyway).  kStackDepth bounds the number of pending siblings; the host builder guarantees
// tree depth + 2 <= kStackDepth (bvh_build.cpp: SAH splits fall back to object-median splits when the remaining depth
// budget is needed), so there is no spill path — the earlier LDS + scratch hybrid cost a generic flat_load per pop.
constexpr int kStackDepth = 32;
constexpr int kLdsStack = kStackDepth;
constexpr int kBlock = 256;
constexpr int32_t kExit = (int32_t)0x80000000;

struct Stack {
    int32_t* lds;                 // &shared[threadIdx.x]
    int top;
    // EXPERIMENT (-DRT_TOPCACHE, off by default): an LDS copy of the first `topCount` nodes of the array (the builder orders nodes by decreasing
    // box area: 64 of the bench tree's 194 k nodes take 32 % of all visits, 256 take 40 %), read with ds_read_b128 instead of vector-L1
    // look-ups.  Measured (profiles/README.md r03): with 64 nodes the two traversal kernels gain 2 %, but the extra branch of every node
    // visit costs 4 % — also when the count is zero — because the kernels are bound by vector-instruction issue, not by the L1.
    const float4* top4 = nullptr; uint32_t topCount = 0;
    RT_DEV void push(int32_t v) { lds[top * kBlock] = v; ++top; }
    RT_DEV int32_t pop() { --top; return lds[top * kBlock]; }
};

// workgroup prologue of a traversal kernel: the LDS copy of the hottest nodes behind the (budget + 1) x kBlock stack entries
RT_DEV const float4* stage_top_nodes(const float4* nodes, uint32_t topCount, int32_t budget, int32_t* s_stack) {
#ifdef RT_TOPCACHE
    float4* dst = reinterpret_cast<float4*>(s_stack + (size_t)(budget + 1) * kBlock);
    for (uint32_t k = threadIdx.x; k < topCount * 4u; k += (uint32_t)kBlock) dst[k] = nodes[k];
    __syncthreads();
    return dst;
#else
    return nullptr;
#endif
}

// One visit of a 4-wide node (layout: rt_host.h).  The child planes are never materialised: with the grid step s_a = 2^(e_a-127)
// and the node origin g, the slab parameter of plane "g_a + q * s_a" is  t = q * (s_a / d_a) + (g_a - o_a) / d_a = fma(q, A_a, B_a),
// A and B computed once per visit (s_a is a power of two, so A is exact), then 24 byte->float conversions and 24 fused
// multiply-adds (issued as 12 v_pk_fma_f32) give the slabs of all four children.  Hit children are ordered by entry distance
// with a 5-comparator network (ties keep slot order), the nearest is visited next and the others are pushed far-to-near.
// Boxes are culled against `cut` (closest hit so far * 1.000001, or the light / visibility distance).
typedef float v2f __attribute__((ext_vector_type(2)));
// (a per-ray -(o/d) would make a node's B = (origin - o)/d one fma instead of sub + mul, but costs three more registers per lane: the
// persistent trace kernels then lose a resident workgroup per CU, which is worth more than the 3 instructions — measured by register count)
struct RayPk { float ox, oy, oz, ix, iy, iz; };
RT_DEV RayPk make_raypk(f3 o, float ix, float iy, float iz) { RayPk r; r.ox = o.x; r.oy = o.y; r.oz = o.z; r.ix = ix; r.iy = iy; r.iz = iz; return r; }
RT_DEV float ubyte_f(uint32_t w, int i) { return (float)((w >> (8 * i)) & 0xFFu); }
// (the entry parameter is clamped to a tiny positive number instead of zero: as the high word of a sort key it must be a normal number, see node_step)
constexpr float kNearClamp = 1e-30f;
// entry / exit parameter of two children (x = first, y = second of the pair) from their near / far plane parameters
// (the exit side is written as the two instructions it should be: through __builtin_fminf the compiler re-canonicalises `cut` — a value it
// cannot see the origin of inside the loop — with an extra v_max_f32 in every visit; the instructions' own NaN rule is minnum's)
RT_DEV float min3_with_cut(float a, float b, float c, float cut) {
    float t, r;
    asm("v_min_f32 %0, %1, %2" : "=v"(t) : "v"(c), "v"(cut));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
RT_DEV void slab_of_pair(v2f nx, v2f fx, v2f ny, v2f fy, v2f nz, v2f fz, float cut, float& n0, float& f0, float& n1, float& f1) {
    n0 = __builtin_fmaxf(__builtin_fmaxf(nx.x, ny.x), __builtin_fmaxf(nz.x, kNearClamp));
    f0 = min3_with_cut(fx.x, fy.x, fz.x, cut);
    n1 = __builtin_fmaxf(__builtin_fmaxf(nx.y, ny.y), __builtin_fmaxf(nz.y, kNearClamp));
    f1 = min3_with_cut(fx.y, fy.y, fz.y, cut);
}
RT_DEV void order_keys(double& a, double& b) {                  // (a, b) <- (min, max): exact on the bit patterns of positive normal doubles
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo; b = hi;
}
// Stack rule: the siblings that are not visited next are pushed one by one (far-to-near) while
//     pending entries + 2 + levels(node) <= budget            (budget = kStackBudget; a test knob can lower it to the level count),
// otherwise ONE "resume" entry (node index and the set of hit slots still to visit) is pushed and the node is fetched and
// tested again when that entry is popped (then against a cut that can only have shrunk).  In resume mode a level costs one
// entry, so by induction pending + levels(node) <= kStackBudget holds at every visit and the 32-entry LDS stack cannot
// overflow for any tree of <= kStackBudget levels (bvh_build.cpp) — without bounding how wide the nodes may be.
constexpr int32_t kResumeBase = 0x40000000;     // references >= this: resume entry = base | node byte offset | mask of the slots still to visit
constexpr uint32_t kNodeOffsetMask = 0x3FFFFFC0u; // an inner reference is the node's byte offset in the array (rt_host.h: device form), < 2^30, 64-byte aligned
constexpr int kStackBudget = kStackDepth - 1;   // one entry is the exit sentinel
// returns the next reference to visit (a child, or the popped stack top when no child is hit)
// COUNT: instrumented variant (child-box tests and node visits per ray; fyprt_set_ray_counting) — the production kernels are
// instantiated without it and carry neither the counters nor their instructions.
template <bool COUNT>
RT_DEV int32_t node_step(const float4* nodes, int32_t budget, int32_t cur, const RayPk& r, float cut, Stack& st, uint32_t& nBox, uint32_t& nNode) {
    // resume entries are rare: one wave-uniform test keeps their decoding (and, below, their slot masks) off the common path — the kernels
    // are bound by instruction issue, every instruction and every exec-mask region of a visit is paid on each of ~18 visits per ray
    const bool anyResumed = __ballot(cur >= kResumeBase) != 0ull;
    const uint32_t off = (uint32_t)cur & kNodeOffsetMask;            // the node's byte offset: one fast-class instruction, resumed or not
    uint32_t allow = 0xFu;
    if (anyResumed) allow = (cur >= kResumeBase) ? ((uint32_t)cur & 0xFu) : 0xFu;
    float4 q0, q1, q2; float2 q3;
#ifdef RT_TOPCACHE
    if ((off >> 6) < st.topCount) {
        const float4* n = st.top4 + (off >> 4);
        q0 = n[0]; q1 = n[1]; q2 = n[2]; q3 = *reinterpret_cast<const float2*>(n + 3);
    } else
#endif
    {
        const float4* n = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nodes) + off);     // uniform base + 32-bit lane offset
        q0 = n[0]; q1 = n[1]; q2 = n[2]; q3 = *reinterpret_cast<const float2*>(n + 3);
    }
    const uint32_t ex = (uint32_t)__float_as_int(q0.w), cnt = (ex >> 24) & 7u, levels = ex >> 27;
    if (COUNT) { nBox += (uint32_t)__popc(allow & ((1u << cnt) - 1u)); nNode += 1u; }
    const float Ax = __int_as_float((int)((ex & 0xFFu) << 23)) * r.ix, Ay = __int_as_float((int)(((ex >> 8) & 0xFFu) << 23)) * r.iy,
                Az = __int_as_float((int)(((ex >> 16) & 0xFFu) << 23)) * r.iz;
    const float Bx = (q0.x - r.ox) * r.ix, By = (q0.y - r.oy) * r.iy, Bz = (q0.z - r.oz) * r.iz;
    const v2f Ax2 = v2f{Ax, Ax}, Ay2 = v2f{Ay, Ay}, Az2 = v2f{Az, Az}, Bx2 = v2f{Bx, Bx}, By2 = v2f{By, By}, Bz2 = v2f{Bz, Bz};
    const uint32_t lx = (uint32_t)__float_as_int(q2.x), ly = (uint32_t)__float_as_int(q2.y), lz = (uint32_t)__float_as_int(q2.z),
                   hx = (uint32_t)__float_as_int(q2.w), hy = (uint32_t)__float_as_int(q3.x), hz = (uint32_t)__float_as_int(q3.y);
    // the ray enters a slab through the lo plane where it travels in +axis direction and through the hi plane otherwise:
    // select the four children's near / far plane bytes with one v_cndmask per axis and side instead of a min and a max per
    // plane pair.  Unused child slots hold lo = 255, hi = 0 on every axis, which this test can never hit.
    const bool bx = r.ix < 0.0f, by = r.iy < 0.0f, bz = r.iz < 0.0f;
    const uint32_t nxq = bx ? hx : lx, fxq = bx ? lx : hx, nyq = by ? hy : ly, fyq = by ? ly : hy, nzq = bz ? hz : lz, fzq = bz ? lz : hz;
    float k0, k1, k2, k3, f0, f1, f2, f3_;
    slab_of_pair(__builtin_elementwise_fma(v2f{ubyte_f(nxq, 0), ubyte_f(nxq, 1)}, Ax2, Bx2), __builtin_elementwise_fma(v2f{ubyte_f(fxq, 0), ubyte_f(fxq, 1)}, Ax2, Bx2),
                 __builtin_elementwise_fma(v2f{ubyte_f(nyq, 0), ubyte_f(nyq, 1)}, Ay2, By2), __builtin_elementwise_fma(v2f{ubyte_f(fyq, 0), ubyte_f(fyq, 1)}, Ay2, By2),
                 __builtin_elementwise_fma(v2f{ubyte_f(nzq, 0), ubyte_f(nzq, 1)}, Az2, Bz2), __builtin_elementwise_fma(v2f{ubyte_f(fzq, 0), ubyte_f(fzq, 1)}, Az2, Bz2),
                 cut, k0, f0, k1, f1);
    slab_of_pair(__builtin_elementwise_fma(v2f{ubyte_f(nxq, 2), ubyte_f(nxq, 3)}, Ax2, Bx2), __builtin_elementwise_fma(v2f{ubyte_f(fxq, 2), ubyte_f(fxq, 3)}, Ax2, Bx2),
                 __builtin_elementwise_fma(v2f{ubyte_f(nyq, 2), ubyte_f(nyq, 3)}, Ay2, By2), __builtin_elementwise_fma(v2f{ubyte_f(fyq, 2), ubyte_f(fyq, 3)}, Ay2, By2),
                 __builtin_elementwise_fma(v2f{ubyte_f(nzq, 2), ubyte_f(nzq, 3)}, Az2, Bz2), __builtin_elementwise_fma(v2f{ubyte_f(fzq, 2), ubyte_f(fzq, 3)}, Az2, Bz2),
                 cut, k2, f2, k3, f3_);
    // Hit children ordered by entry distance: the four (entry distance, child reference) pairs are sorted as 64-bit keys — the distance's
    // bit pattern in the high word, the reference in the low word — with v_min_f64 / v_max_f64: for positive normal doubles the numeric
    // order IS the order of the bit patterns, every distance here is >= 1e-30 (the near clamp of slab_of_pair) and no larger than +inf, which
    // as a high word gives exponent fields between 0x0DA and 0x7F8: normal, finite.  A comparator is two 4-cycle instructions instead of a
    // compare and four selects (profiles/r03/microbench.jsonl: v_min_f64 4.3 cycles, v_cmp + v_cndmask 4.1 each).  Equal distances
    // are ordered by the reference (any fixed rule does: the oracle's twin applies the same one).
    constexpr uint32_t kMissHi = 0x7F900000u;                         // above every distance (+inf = 0x7F800000), still a finite double
    uint32_t h0 = (k0 <= f0) ? __float_as_uint(k0) : kMissHi;
    uint32_t h1 = (k1 <= f1) ? __float_as_uint(k1) : kMissHi;
    uint32_t h2 = (k2 <= f2) ? __float_as_uint(k2) : kMissHi;
    uint32_t h3 = (k3 <= f3_) ? __float_as_uint(k3) : kMissHi;
    if (anyResumed) {                                                 // rare: a resumed visit only looks at the slots still owed
        h0 = (allow & 1u) ? h0 : kMissHi; h1 = (allow & 2u) ? h1 : kMissHi; h2 = (allow & 4u) ? h2 : kMissHi; h3 = (allow & 8u) ? h3 : kMissHi;
    }
    const int32_t c0 = __float_as_int(q1.x), c1 = __float_as_int(q1.y), c2 = __float_as_int(q1.z), c3 = __float_as_int(q1.w);
#ifdef RT_SORT_CNDMASK      // timing experiment only (compare + four selects per comparator, ties in slot order: NOT the oracle twin's order)
    uint32_t s0h = h0, s1h = h1, s2h = h2, s3h = h3; int32_t r0 = c0, r1 = c1, r2 = c2, r3 = c3;
    auto cs = [](uint32_t& ka, int32_t& ra, uint32_t& kb, int32_t& rb) { const bool sw = kb < ka; const uint32_t k0 = sw ? kb : ka, k1 = sw ? ka : kb; const int32_t x0 = sw ? rb : ra, x1 = sw ? ra : rb; ka = k0; kb = k1; ra = x0; rb = x1; };
    cs(s0h, r0, s1h, r1); cs(s2h, r2, s3h, r3); cs(s0h, r0, s2h, r2); cs(s1h, r1, s3h, r3); cs(s1h, r1, s2h, r2);
#else
    double d0 = __hiloint2double((int)h0, c0), d1 = __hiloint2double((int)h1, c1), d2 = __hiloint2double((int)h2, c2), d3 = __hiloint2double((int)h3, c3);
    order_keys(d0, d1); order_keys(d2, d3); order_keys(d0, d2); order_keys(d1, d3); order_keys(d1, d2);
    const uint32_t s0h = (uint32_t)__double2hiint(d0), s1h = (uint32_t)__double2hiint(d1), s2h = (uint32_t)__double2hiint(d2), s3h = (uint32_t)__double2hiint(d3);
    const int32_t r0 = __double2loint(d0), r1 = __double2loint(d1), r2 = __double2loint(d2), r3 = __double2loint(d3);
#endif
    // unconditional stores, conditional advance: three ds_write_b32 without a branch each (entries top .. top + 2 exist: the
    // rule above leaves room for them; a slot written for a child that was not hit is simply overwritten by the next push)
#define RT_PUSH3() do { st.lds[st.top * kBlock] = r3; st.top += (s3h < kMissHi) ? 1 : 0; \
                        st.lds[st.top * kBlock] = r2; st.top += (s2h < kMissHi) ? 1 : 0; \
                        st.lds[st.top * kBlock] = r1; st.top += (s1h < kMissHi) ? 1 : 0; } while (0)
    const uint64_t noRoom = __ballot((st.top - 1) + 2 + (int)levels > budget);
    if (noRoom == 0ull) RT_PUSH3();                                   // the common case, wave-uniform: no exec-mask region around the pushes
    else if (((noRoom >> (threadIdx.x & 63u)) & 1ull) == 0ull) RT_PUSH3();
    else if (s1h < kMissHi) {                                         // two or more hits and no room to push them one by one
        const uint32_t hit = (h0 < kMissHi ? 1u : 0u) | (h1 < kMissHi ? 2u : 0u) | (h2 < kMissHi ? 4u : 0u) | (h3 < kMissHi ? 8u : 0u);
        const uint32_t nearest = (c0 == r0) ? 1u : (c1 == r0) ? 2u : (c2 == r0) ? 4u : 8u;     // the nearest child's slot (references of hit slots are distinct)
        st.push(kResumeBase | (int32_t)(off | (hit & ~nearest)));
    }
#undef RT_PUSH3
    return (s0h < kMissHi) ? r0 : st.pop();
}

// A ray with a NaN or infinite component can hit no triangle (every Möller–Trumbore comparison fails), but its slab tests
// would pass everywhere: such rays are answered as misses up front instead of walking the whole tree.
RT_DEV bool ray_not_finite(f3 o, f3 d) {
    const float z = ((o.x - o.x) + (o.y - o.y)) + ((o.z - o.z) + (d.x - d.x)) + ((d.y - d.y) + (d.z - d.z));
    return !(z == 0.0f);
}
RT_DEV float safe_inv(float d) { return 1.0f / ((__builtin_fabsf(d) < 1e-30f) ? __builtin_copysignf(1e-30f, d) : d); }

// Möller–Trumbore with exactly the reference's operation order (Renderer.cu:513-537) on a leaf record (v0, e1 = v1-v0, e2 = v2-v0):
// true and (t, u, v) if the ray passes through the triangle at t > 1e-4 (the caller compares t with its own interval), no culling.
RT_DEV bool tri_test(const float4* tp, f3 o, f3 d, float& t, float& u, float& v, uint32_t& id) {
    const float4 a = tp[0], b = tp[1], c = tp[2];
    id = (uint32_t)__float_as_int(c.y);
    const f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
    const f3 hh = cross(d, e2);
    const float det = dot(e1, hh), f = 1.0f / det;
    const f3 s = o - v0;
    u = f * dot(s, hh);
    if (u < 0.0f || u > 1.0f) return false;
    const f3 q = cross(s, e1);
    v = f * dot(d, q);
    if (v < 0.0f || (u + v) > 1.0f) return false;
    t = f * dot(e2, q);
    return t > 0.0001f;
}
// the same test against an ORIGINAL triangle (triPos record): the light triangle of a shadow ray; -1 if it is not hit
RT_DEV float light_tri_distance(const DevScene& sc, uint32_t tri, f3 o, f3 d) {
    const float4* p = sc.triPos + (size_t)tri * 3;
    const float4 a = p[0], b = p[1], c = p[2];
    const f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(b.x, b.y, b.z) - v0, e2 = mk3(c.x, c.y, c.z) - v0;
    const f3 hh = cross(d, e2);
    const float det = dot(e1, hh), f = 1.0f / det;
    const f3 s = o - v0;
    const float u = f * dot(s, hh);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const f3 q = cross(s, e1);
    const float v = f * dot(d, q);
    if (v < 0.0f || (u + v) > 1.0f) return -1.0f;
    const float t = f * dot(e2, q);
    return (t > 0.0001f) ? t : -1.0f;
}

// Closest hit over the acceleration structure (ONE tree: the reference's TLAS and per-mesh BLAS trees are merged at build).
// Ordered traversal (nearest hit child first, the others pushed far-to-near), boxes culled against closest * 1.000001f; a
// triangle is accepted when 1e-4 < t < closest, no back-face culling.
// The node-loop quorum is given for a full wave; a wave in which only some lanes still have a ray (a thread-per-ray kernel late in its life, a
// persistent wave before its refill) scales it to those lanes — with a fixed count such a wave would leave the loop after every single visit.
RT_DEV uint32_t quorum_of(uint32_t q, uint32_t lanesWithRay) { return (q * lanesWithRay + 63u) >> 6; }

template <bool COUNT>
RT_DEV Hit trace_closest(const DevScene& sc, f3 o, f3 d, int32_t* ldsBase, const float4* top4 = nullptr) {
    Hit h; h.t = 3.402823466e+38f; h.u = 0.0f; h.v = 0.0f; h.tri = -1;
    uint32_t nBox = 0, nTri = 0, nNode = 0;
    if (sc.triCount == 0 || ray_not_finite(o, d)) { if (COUNT) atomicAdd(sc.rayCounter, 1ull); return h; }
    const RayPk pk = make_raypk(o, safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    float closestInfl = h.t * 1.000001f;
    Stack st; st.lds = ldsBase; st.top = 0; st.top4 = top4; st.topCount = top4 ? sc.topCount : 0u; st.push(kExit);
    int32_t cur = sc.rootRef;
    while (true) {
        bool walk = cur >= 0;                           // (one compare per round serves the loop condition and the quorum ballot)
        const uint32_t quorum = quorum_of(sc.nodeQuorum, (uint32_t)__popcll(__ballot(true)));     // lanes still in this loop = lanes with a ray
        while (walk) {
            cur = node_step<COUNT>(sc.nodes, sc.stackBudget, cur, pk, closestInfl, st, nBox, nNode);
            // lanes that reached a leaf wait outside this loop; once only a few lanes are still walking inner nodes,
            // stop and let everybody test their leaves (keeps SIMD lanes busy; pure scheduling, results unchanged)
            walk = cur >= 0;
            if ((uint32_t)__popcll(__ballot(walk)) < quorum) break;
        }
        if (cur >= 0) continue;
        if (cur == kExit) break;
        const uint32_t code = (uint32_t)~cur, first = code >> 2, cnt = (code & 3u) + 1u;
        for (uint32_t k = 0; k < cnt; ++k) {
            float t, u, v; uint32_t id;
            if (COUNT) nTri += 1;
            if (tri_test(sc.leafTris + (size_t)(first + k) * 3, o, d, t, u, v, id) && t < h.t) { h.t = t; h.u = u; h.v = v; h.tri = (int32_t)id; closestInfl = t * 1.000001f; }
        }
        cur = st.pop();
    }
    if (COUNT) {     // SURVEY.md §8(d) instrumentation; same counts as the oracle's restatement (tests/test_gpu_counters.py)
        atomicAdd(sc.rayCounter + 0, 1ull); atomicAdd(sc.rayCounter + 1, (unsigned long long)nBox);
        atomicAdd(sc.rayCounter + 2, (unsigned long long)nTri); atomicAdd(sc.rayCounter + 3, (unsigned long long)(h.tri >= 0 ? 1 : 0));
        atomicAdd(sc.rayCounter + 4, (unsigned long long)nNode);
    }
    return h;
}

// Visibility query towards a known light triangle (shadow rays of ReSTIR DI Part 2, NEE and
// light-source sampling).  The reference runs a full closest-hit TraceRay and then compares the hit
// index with the light (R.cu:2014-2031, :1383-1403, :1501-1504); all the callers use of the result is
//   (objectIndex == light && hitDistance >= 0) | hitDistance < 0 (nothing hit at all) | anything else.
// So: intersect the light triangle first (same Möller–Trumbore), then traverse with the interval cut at
// t_light and stop at the FIRST triangle closer than the light (any-hit).  If the ray misses the light
// triangle itself (edge rounding) fall back to the full closest-hit query.  Exact-t ties count as
// "not closer" (the reference's own tie order is traversal-order dependent).
struct ShadowHit { float hitDistance; int32_t objectIndex; };
template <bool COUNT>
RT_DEV ShadowHit trace_shadow(const DevScene& sc, f3 o, f3 d, uint32_t lightTri, int32_t* ldsBase, const float4* top4 = nullptr) {
    ShadowHit r;
    const float tL = light_tri_distance(sc, lightTri, o, d);
    if (!(tL > 0.0f)) {                               // light not hit by its own shadow ray: exact fallback
        if (COUNT) atomicAdd(sc.rayCounter + 2, 1ull);
        const Hit h = trace_closest<COUNT>(sc, o, d, ldsBase, top4);
        r.hitDistance = (h.tri < 0) ? -1.0f : h.t; r.objectIndex = h.tri;
        return r;
    }
    uint32_t nBox = 0, nTri = 1, nNode = 0;
    const RayPk pk = make_raypk(o, safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    const float cut = tL * 1.000001f;
    Stack st; st.lds = ldsBase; st.top = 0; st.top4 = top4; st.topCount = top4 ? sc.topCount : 0u; st.push(kExit);
    int32_t cur = sc.rootRef;
    r.hitDistance = tL; r.objectIndex = (int32_t)lightTri;
    bool occluded = false;
    while (!occluded) {
        bool walk = cur >= 0;
        const uint32_t quorum = quorum_of(sc.nodeQuorum, (uint32_t)__popcll(__ballot(true)));
        while (walk) {
            cur = node_step<COUNT>(sc.nodes, sc.stackBudget, cur, pk, cut, st, nBox, nNode);
            walk = cur >= 0;
            if ((uint32_t)__popcll(__ballot(walk)) < quorum) break;
        }
        if (cur >= 0) continue;
        if (cur == kExit) break;
        const uint32_t code = (uint32_t)~cur, first = code >> 2, cnt = (code & 3u) + 1u;
        for (uint32_t k = 0; k < cnt; ++k) {
            const float4* tp = sc.leafTris + (size_t)(first + k) * 3;
            if ((uint32_t)__float_as_int(tp[2].y) == lightTri) continue;
            float t, u, v; uint32_t id;
            if (COUNT) nTri += 1;
            if (tri_test(tp, o, d, t, u, v, id) && t < tL) { r.hitDistance = t; r.objectIndex = (int32_t)id; occluded = true; break; }
        }
        cur = st.pop();
    }
    if (COUNT) {
        atomicAdd(sc.rayCounter + 0, 1ull); atomicAdd(sc.rayCounter + 1, (unsigned long long)nBox);
        atomicAdd(sc.rayCounter + 2, (unsigned long long)nTri); atomicAdd(sc.rayCounter + 3, 1ull);
        atomicAdd(sc.rayCounter + 4, (unsigned long long)nNode);
    }
    return r;
}

// Miss (Renderer.cu:2423-2429; worldPosition / objectIndex zero-filled / -1: DESIGN.md §5 R1)
RT_DEV Payload make_miss() { Payload p; p.hitDistance = -1.0f; p.px = p.py = p.pz = 0.0f; p.nx = p.ny = p.nz = 0.0f; p.u = 0.0f; p.v = 0.0f; p.objectIndex = -1; return p; }
// ClosestHit (Renderer.cu:2389-2421) from the per-triangle shading record
RT_DEV Payload make_hit(const DevScene& sc, f3 o, f3 d, const Hit& h) {
    const float4* s = sc.triShade + (size_t)h.tri * 4;
    const float4 s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3];
    Payload p; p.hitDistance = h.t; p.objectIndex = h.tri;
    const f3 pos = o + d * h.t;
    p.px = pos.x; p.py = pos.y; p.pz = pos.z;
    const float w = (1.0f - h.u) - h.v;
    const f3 n = normalize((mk3(s0.x, s0.y, s0.z) * w + mk3(s1.x, s1.y, s1.z) * h.u) + mk3(s2.x, s2.y, s2.z) * h.v);
    p.nx = n.x; p.ny = n.y; p.nz = n.z;
    p.u = (s0.w * w + s2.w * h.u) + s3.y * h.v;      // uv0.x, uv1.x, uv2.x
    p.v = (s1.w * w + s3.x * h.u) + s3.z * h.v;      // uv0.y, uv1.y, uv2.y
    return p;
}
template <bool COUNT>
RT_DEV Payload trace_ray(const DevScene& sc, f3 o, f3 d, int32_t* ldsBase, const float4* top4 = nullptr) {
    const Hit h = trace_closest<COUNT>(sc, o, d, ldsBase, top4);
    return (h.tri < 0) ? make_miss() : make_hit(sc, o, d, h);
}

// ---------------------------------------------------------------------------------------------
struct Mat { f3 albedo; uint32_t useMap, mapIndex; float roughness, metallic, power; f3 emColor; };
RT_DEV Mat load_mat(const DevScene& sc, int idx) {
    const float4* m = sc.mats + (size_t)idx * 3;
    const float4 a = m[0], b = m[1], c = m[2];
    Mat r; r.albedo = mk3(a.x, a.y, a.z);
    const uint32_t info = (uint32_t)__float_as_int(a.w); r.useMap = info >> 31; r.mapIndex = info & 0x7FFFFFFFu;
    r.roughness = b.x; r.metallic = b.y; r.power = b.z; r.emColor = mk3(c.x, c.y, c.z);
    return r;
}
RT_DEV int tri_material(const DevScene& sc, int tri) { return __float_as_int(sc.triShade[(size_t)tri * 4 + 3].w); }
RT_DEV f3 emission(const Mat& m) { return m.emColor * m.power; }                   // Material.cu:5-8

// Texture::SampleBilinear (Texture.cu:103-139) -> RGBA8 re-quantised -> unpacked by the caller
RT_DEV f3 sample_albedo(const DevScene& sc, const Mat& m, float u, float v) {
    if (m.useMap && m.mapIndex < sc.textureCount) {
        const DevTexture t = sc.textures[m.mapIndex];
        u = (u < 0.0f) ? 0.0f : (u > 1.0f ? 1.0f : u);
        v = (v < 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
        const float x = u * (float)(t.width - 1), y = v * (float)(t.height - 1);
        const int x0 = (int)x, y0 = (int)y;
        const int x1 = (x0 + 1 < (int)t.width) ? x0 + 1 : x0, y1 = (y0 + 1 < (int)t.height) ? y0 + 1 : y0;
        const float tx = x - (float)x0, ty = y - (float)y0;
        const f4 c00 = unpack_abgr(t.pixels[y0 * t.width + x0]), c10 = unpack_abgr(t.pixels[y0 * t.width + x1]);
        const f4 c01 = unpack_abgr(t.pixels[y1 * t.width + x0]), c11 = unpack_abgr(t.pixels[y1 * t.width + x1]);
        const f4 cx0 = c00 * (1.0f - tx) + c10 * tx, cx1 = c01 * (1.0f - tx) + c11 * tx;
        const f4 c = unpack_abgr(pack_abgr(cx0 * (1.0f - ty) + cx1 * ty));
        return mk3(c.x, c.y, c.z);
    }
    return m.albedo;
}

// Camera::RecalculateRayDirections (Camera.cpp:136-153) for one pixel
RT_DEV f3 ray_direction(const DevCamera& cam, uint32_t x, uint32_t y) {
    const float cx = ((float)x / (float)cam.W) * 2.0f - 1.0f, cy = ((float)y / (float)cam.H) * 2.0f - 1.0f;
    const f4 target = mul(cam.invProj, mk4(cx, cy, 1.0f, 1.0f));
    const f3 dl = normalize(mk3(target.x, target.y, target.z) / target.w);
    const f4 w = mul(cam.invView, mk4(dl.x, dl.y, dl.z, 0.0f));
    return mk3(w.x, w.y, w.z);
}

// Triangle helpers (Triangle.cuh:14-59) on a triPos record
struct TriGeom { f3 p0, p1, p2, n0, n1, n2; int mat; };
RT_DEV TriGeom load_tri(const DevScene& sc, uint32_t tri) {
    const float4* p = sc.triPos + (size_t)tri * 3; const float4* s = sc.triShade + (size_t)tri * 4;
    const float4 a = p[0], b = p[1], c = p[2], s0 = s[0], s1 = s[1], s2 = s[2];
    TriGeom g; g.p0 = mk3(a.x, a.y, a.z); g.p1 = mk3(b.x, b.y, b.z); g.p2 = mk3(c.x, c.y, c.z);
    g.n0 = mk3(s0.x, s0.y, s0.z); g.n1 = mk3(s1.x, s1.y, s1.z); g.n2 = mk3(s2.x, s2.y, s2.z); g.mat = __float_as_int(a.w);
    return g;
}
RT_DEV f3 tri_centroid(const TriGeom& g) { return ((g.p0 + g.p1) + g.p2) / 3.0f; }
RT_DEV f3 tri_normal(const TriGeom& g) { return normalize(((g.n0 + g.n1) + g.n2) / 3.0f); }
RT_DEV float tri_area(const TriGeom& g) { return 0.5f * length(cross(g.p1 - g.p0, g.p2 - g.p0)); }
RT_DEV f3 tri_random_point(const TriGeom& g, uint32_t& seed) {
    const float r1 = rnd(seed), r2 = rnd(seed), s = __builtin_sqrtf(r1);
    const float u = 1.0f - s, v = (1.0f - r2) * s, w = r2 * s;
    return (u * g.p0 + v * g.p1) + w * g.p2;
}

}  // namespace rt
