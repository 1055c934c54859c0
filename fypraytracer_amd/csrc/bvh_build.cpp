// Host builder of the library's own acceleration structure (layout: rt_host.h).
// Binned SAH (16 bins, 3 axes) over triangle centroids, up to 4 triangles per leaf, two levels (a tree per mesh under a
// tree over the meshes) merged into one binary tree, which is then collapsed into 4-wide nodes with 8-bit quantised child
// boxes (one 64-byte fetch per four children), nodes in pre-order so the top of the tree is contiguous in memory.  Depth is bounded BY CONSTRUCTION: every
// subtree gets a depth budget (the bound for the TLAS root, the remainder for each BLAS) and a node whose
// remaining budget is only just enough for a balanced subtree of its size is split at the object median
// instead of the SAH plane, so leaf depth stays bounded (see kMaxDepthRelaxed / kMaxDepthSafe below).  This replaces the reference's recursive builder
// (BVH.cpp:146-309, one triangle per leaf, unordered) — the tree SHAPE is ours; the set of
// triangles a ray can reach is the same, and the closest hit is found with the reference's
// own Möller–Trumbore arithmetic, so results match except for exact-tie order.
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <queue>
#include "rt_host.h"

namespace rth {
namespace {

struct Box {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float* l, const float* h) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); } }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    float area() const { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return (dx < 0) ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx); }
};
struct Prim { Box b; float c[3]; uint32_t id; };

// Leaf depth bound of the binary tree.  What the kernels need is a WIDE tree of at most kStackBudget (31) levels (node_step's
// stack rule); a wide level swallows about two binary ones, so the binary tree is first built with a generous bound (SAH
// splits almost everywhere) and only if its collapse ends up deeper than 31 levels — pathological input — once more with the
// bound that guarantees it (wide levels <= binary height <= 30).
constexpr uint32_t kMaxDepthRelaxed = 48, kMaxDepthSafe = 30;
inline uint32_t ceilLog2(uint32_t n) { uint32_t l = 0; while ((1u << l) < n) ++l; return l; }

// binary node of the intermediate SAH tree (both child boxes in the parent)
struct Node2 { float lo0[3], hi0[3], lo1[3], hi1[3]; int32_t child0, child1; };

struct Builder {
    uint64_t forced = 0;
    std::vector<Node2> nodes; uint32_t maxDepth = 0; uint32_t maxLeaf; uint32_t depthLimit = kMaxDepthSafe;
    std::function<int32_t(const Prim*, uint32_t, uint32_t)> makeLeaf;   // (prims, count, depth) -> leaf ref
    static constexpr int kMaxBins = 64;
    int kBins = 16;                     // experiments: FYPRT_BVH_BINS
    uint32_t sweepBelow = 0;            // nodes of fewer primitives than this try every split position (exact sweep SAH): FYPRT_BVH_SWEEP

    float nodeCost = 1.0f;              // SAH: cost of visiting a node relative to testing a triangle (experiments: FYPRT_BVH_NODE_COST)

    int32_t build(Prim* p, uint32_t first, uint32_t last, uint32_t depth, Box& outBox) {
        const uint32_t count = last - first;
        Box nb, cb;
        for (uint32_t i = first; i < last; ++i) { nb.grow(p[i].b); cb.grow(p[i].c, p[i].c); }
        outBox = nb;
        int bestAxis = -1, bestBin = -1; float bestCost = FLT_MAX;
        // leaves needed below this node if split evenly from here on: ceil(count / maxLeaf) -> levels = ceilLog2(that)
        const uint32_t balancedLevels = ceilLog2((count + maxLeaf - 1) / maxLeaf);
        const bool mustBalance = depth + balancedLevels + 1 >= depthLimit;
        if (mustBalance && count > maxLeaf) ++forced;
        bool swept = false;
        if (count > 1 && !mustBalance && count < sweepBelow) {
            std::vector<float> ra(count);
            for (int axis = 0; axis < 3; ++axis) {
                if (!(cb.hi[axis] > cb.lo[axis])) continue;
                std::sort(p + first, p + last, [axis](const Prim& a, const Prim& b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.id < b.id); });
                Box acc;
                for (uint32_t i = count - 1; i > 0; --i) { acc.grow(p[first + i].b); ra[i] = acc.area(); }
                acc = Box();
                for (uint32_t i = 0; i + 1 < count; ++i) {
                    acc.grow(p[first + i].b);
                    const float cost = acc.area() * (float)(i + 1) + ra[i + 1] * (float)(count - i - 1);
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = (int)(i + 1); }
                }
            }
            swept = bestAxis >= 0;
        }
        if (count > 1 && !mustBalance && !swept) {
            for (int axis = 0; axis < 3; ++axis) {
                const float cmin = cb.lo[axis], cmax = cb.hi[axis];
                if (!(cmax > cmin)) continue;
                Box bb[kMaxBins]; uint32_t bc[kMaxBins] = {0};
                const float scale = (float)kBins / (cmax - cmin);
                for (uint32_t i = first; i < last; ++i) {
                    int b = std::min(kBins - 1, std::max(0, (int)((p[i].c[axis] - cmin) * scale)));
                    bc[b]++; bb[b].grow(p[i].b);
                }
                float rightArea[kMaxBins]; uint32_t rightCount[kMaxBins]; Box acc; uint32_t n = 0;
                for (int b = kBins - 1; b > 0; --b) { acc.grow(bb[b]); n += bc[b]; rightArea[b] = acc.area(); rightCount[b] = n; }
                acc = Box(); n = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bb[b]); n += bc[b];
                    if (n == 0 || rightCount[b + 1] == 0) continue;
                    float cost = acc.area() * (float)n + rightArea[b + 1] * (float)rightCount[b + 1];
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
                }
            }
        }
        if (count <= maxLeaf && mustBalance) { maxDepth = std::max(maxDepth, depth); return makeLeaf(p + first, count, depth); }
        if (count <= maxLeaf) {
            const float leafCost = (float)count * nb.area();
            const float splitCost = (bestAxis >= 0) ? nodeCost * nb.area() + bestCost : FLT_MAX;
            if (count == 1 || leafCost <= splitCost) { maxDepth = std::max(maxDepth, depth); return makeLeaf(p + first, count, depth); }
        }
        uint32_t mid;
        if (swept) {
            const int axis = bestAxis;
            std::sort(p + first, p + last, [axis](const Prim& a, const Prim& b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.id < b.id); });
            mid = first + (uint32_t)bestBin;
        } else if (bestAxis >= 0) {
            const float cmin = cb.lo[bestAxis], scale = (float)kBins / (cb.hi[bestAxis] - cmin);
            Prim* m = std::partition(p + first, p + last, [&](const Prim& q) {
                return std::min(kBins - 1, std::max(0, (int)((q.c[bestAxis] - cmin) * scale))) <= bestBin; });
            mid = (uint32_t)(m - p);
        } else {
            int axis = 0; float ext = -1.0f;
            for (int a = 0; a < 3; ++a) { float e = cb.hi[a] - cb.lo[a]; if (e > ext) { ext = e; axis = a; } }
            mid = first + count / 2;
            std::nth_element(p + first, p + mid, p + last, [axis](const Prim& a, const Prim& b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.id < b.id); });
        }
        if (mid == first || mid == last) mid = first + count / 2;
        const int32_t self = (int32_t)nodes.size();
        nodes.emplace_back();
        Box b0, b1;
        int32_t c0 = build(p, first, mid, depth + 1, b0);
        int32_t c1 = build(p, mid, last, depth + 1, b1);
        Node2& n = nodes[self];
        std::memcpy(n.lo0, b0.lo, 12); std::memcpy(n.hi0, b0.hi, 12); std::memcpy(n.lo1, b1.lo, 12); std::memcpy(n.hi1, b1.hi, 12);
        n.child0 = c0; n.child1 = c1;
        return self;
    }
};


// ---- collapse of the binary tree into 4-wide nodes ------------------------------------------------------------
// Which descendants become the (up to four) children of a wide node is decided by a dynamic programme that minimises the
// summed surface area of the wide nodes (below).  Every node records how many wide levels its subtree has: the traversal
// pushes the siblings it does not visit next one by one while  pending + 2 + levels <= kStackBudget  and otherwise a
// single "resume this node" entry (rt_device.h: node_step), so the pending-entry count can never exceed
// kStackBudget as long as the tree has at most kStackBudget levels — which BuildSceneBVH guarantees.
struct Collapser {
    const std::vector<Node2>& bn; std::vector<uint8_t> height; std::vector<Node> out; std::vector<float> outArea;   // outArea: surface area of each wide node's box (visit probability of a random ray)
    // SAH-optimal collapse (dynamic programme over the binary tree): cost[n][j-1] = least sum of wide-node surface areas that
    // covers the subtree of binary node n with at most j roots (a root is a wide node or a leaf reference); sel[n][j-1] = how
    // the j roots are dealt to the two children ({0,0} = "n itself is the one root").  The expected number of node visits of a
    // random ray is proportional to that sum and the leaves are fixed, so this is the cheapest 4-wide tree obtainable from
    // the binary one.
    struct Child { int32_t ref; Box b; };
    struct Sel { uint8_t j0, j1; };
    std::vector<std::array<float, 4>> cost; std::vector<std::array<Sel, 4>> sel;
    explicit Collapser(const std::vector<Node2>& b) : bn(b), height(b.size(), 0), cost(b.size()), sel(b.size()) {}
    float T(int32_t ref, int j) const { return ref < 0 ? 0.0f : cost[(size_t)ref][j - 1]; }
    void solve(int32_t ref) {
        if (ref < 0) return;
        const Node2& b = bn[ref];
        solve(b.child0); solve(b.child1);
        Box nb; nb.grow(b.lo0, b.hi0); nb.grow(b.lo1, b.hi1);
        auto& c = cost[(size_t)ref]; auto& sp = sel[(size_t)ref];
        float best[5]; Sel arg[5];
        for (int j = 2; j <= 4; ++j) {
            best[j] = FLT_MAX; arg[j] = Sel{1, (uint8_t)(j - 1)};
            for (int j0 = 1; j0 < j; ++j0) { const float v = T(b.child0, j0) + T(b.child1, j - j0); if (v < best[j]) { best[j] = v; arg[j] = Sel{(uint8_t)j0, (uint8_t)(j - j0)}; } }
        }
        c[0] = nb.area() + best[4]; sp[0] = Sel{0, 0};
        for (int j = 2; j <= 4; ++j) {
            if (c[j - 2] <= best[j]) { c[j - 1] = c[j - 2]; sp[j - 1] = sp[j - 2]; }      // "at most j": fewer roots are allowed
            else { c[j - 1] = best[j]; sp[j - 1] = arg[j]; }
        }
    }
    // the roots the programme chooses for (ref, at most j roots), appended to ch[] with their boxes
    void roots(int32_t ref, const Box& box, int j, Child* ch, int& k) const {
        if (ref < 0 || sel[(size_t)ref][j - 1].j0 == 0) { ch[k].ref = ref; ch[k].b = box; ++k; return; }
        const Sel s = sel[(size_t)ref][j - 1];
        const Node2& b = bn[ref]; Box b0, b1; std::memcpy(b0.lo, b.lo0, 12); std::memcpy(b0.hi, b.hi0, 12); std::memcpy(b1.lo, b.lo1, 12); std::memcpy(b1.hi, b.hi1, 12);
        roots(b.child0, b0, s.j0, ch, k); roots(b.child1, b1, s.j1, ch, k);
    }
    uint32_t h(int32_t ref) const { return ref < 0 ? 0u : height[(size_t)ref]; }
    uint32_t computeHeights(int32_t ref) {
        if (ref < 0) return 0;
        const uint32_t v = 1u + std::max(computeHeights(bn[ref].child0), computeHeights(bn[ref].child1));
        height[(size_t)ref] = (uint8_t)v; return v;
    }
    static void quantise(Node& n, const Child* ch, int k) {
        Box nb; for (int i = 0; i < k; ++i) nb.grow(ch[i].b);
        std::memset(&n, 0, sizeof n);
        for (int a = 0; a < 3; ++a) for (int i = k; i < 4; ++i) { n.qlo[a][i] = 255; n.qhi[a][i] = 0; }   // unused slots: an inverted box no ray can hit (node_step)
        n.meta = (uint8_t)k;
        for (int a = 0; a < 3; ++a) {
            const float lo = nb.lo[a]; n.origin[a] = lo;
            const double ext = (double)nb.hi[a] - (double)lo;
            int e = 1;
            if (ext > 0.0) { int ee; const double m = std::frexp(ext / 254.0, &ee); if (m == 0.5) --ee; e = std::min(254, std::max(1, ee + 127)); }   // 2^(e-127) >= ext/254
            for (;; ++e) {
                const double s = std::ldexp(1.0, e - 127); const float sf = (float)s;
                bool ok = ext <= 254.0 * s;
                for (int i = 0; i < k && ok; ++i) {
                    int ql = (int)std::floor(((double)ch[i].b.lo[a] - (double)lo) / s - 0.0625), qh = (int)std::ceil(((double)ch[i].b.hi[a] - (double)lo) / s + 0.0625);
                    ql = std::min(255, std::max(0, ql)); qh = std::min(255, std::max(0, qh));
                    while (ql > 0 && !(std::fmaf((float)ql, sf, lo) <= ch[i].b.lo[a])) --ql;
                    while (qh < 255 && !(std::fmaf((float)qh, sf, lo) >= ch[i].b.hi[a])) ++qh;
                    ok = std::fmaf((float)ql, sf, lo) <= ch[i].b.lo[a] && std::fmaf((float)qh, sf, lo) >= ch[i].b.hi[a];
                    n.qlo[a][i] = (uint8_t)ql; n.qhi[a][i] = (uint8_t)qh;
                }
                if (ok || e >= 254) break;
            }
            n.ex[a] = (uint8_t)e;
        }
    }
    // returns the wide index; `levels` = wide levels of the subtree rooted here (1 = all children are leaves)
    int32_t emit(int32_t ref, uint32_t& levels) {
        Child ch[4]; int k = 0;
        const Node2& bnode = bn[ref];
        Box b0, b1; std::memcpy(b0.lo, bnode.lo0, 12); std::memcpy(b0.hi, bnode.hi0, 12); std::memcpy(b1.lo, bnode.lo1, 12); std::memcpy(b1.hi, bnode.hi1, 12);
        float best = FLT_MAX; int bj0 = 1;
        for (int j0 = 1; j0 < 4; ++j0) { const float v = T(bnode.child0, j0) + T(bnode.child1, 4 - j0); if (v < best) { best = v; bj0 = j0; } }
        roots(bnode.child0, b0, bj0, ch, k); roots(bnode.child1, b1, 4 - bj0, ch, k);
        const int32_t self = (int32_t)out.size();
        out.emplace_back(); outArea.push_back(0.0f);
        if (self == 0) { Box nb; for (int i = 0; i < k; ++i) nb.grow(ch[i].b); outArea[0] = nb.area(); }
        Node n; quantise(n, ch, k);
        uint32_t below = 0;
        for (int i = 0; i < k; ++i) {
            if (ch[i].ref >= 0) { uint32_t cl = 0; n.child[i] = emit(ch[i].ref, cl); outArea[(size_t)n.child[i]] = ch[i].b.area(); below = std::max(below, cl); }
            else n.child[i] = ch[i].ref;
        }
        for (int i = k; i < 4; ++i) n.child[i] = INT32_MIN;                       // never read: the count masks the slot
        levels = 1u + below;
        n.meta = (uint8_t)((uint32_t)k | (levels << 3));
        out[(size_t)self] = n;
        return self;
    }
};

// Debug aid (FYPRT_BVH_DEBUG): the dynamic programme's objective — summed surface area of the wide nodes, proportional to the expected
// number of node visits of a random ray — for a collapse into nodes of up to WIDTH children (what a wider node format would visit).
template <int WIDTH> double collapseObjective(const std::vector<Node2>& bn, int32_t root) {
    std::vector<std::array<float, WIDTH>> cost(bn.size());
    std::function<void(int32_t)> solve = [&](int32_t ref) {
        if (ref < 0) return;
        const Node2& b = bn[ref];
        solve(b.child0); solve(b.child1);
        Box nb; nb.grow(b.lo0, b.hi0); nb.grow(b.lo1, b.hi1);
        auto T = [&](int32_t r, int j) { return r < 0 ? 0.0f : cost[(size_t)r][j - 1]; };
        float best[WIDTH + 1];
        for (int j = 2; j <= WIDTH; ++j) { best[j] = FLT_MAX; for (int j0 = 1; j0 < j; ++j0) best[j] = std::min(best[j], T(b.child0, j0) + T(b.child1, j - j0)); }
        auto& c = cost[(size_t)ref];
        c[0] = nb.area() + best[WIDTH];
        for (int j = 2; j <= WIDTH; ++j) c[j - 1] = std::min(c[j - 2], best[j]);
    };
    solve(root);
    return root < 0 ? 0.0 : (double)cost[(size_t)root][0];
}

inline const uint32_t* triIdx(const uint8_t* tris, uint32_t stride, uint32_t i) { return reinterpret_cast<const uint32_t*>(tris + (size_t)i * stride); }

}  // namespace

static void BuildWithDepthBound(const fyprt_vertex* verts, const uint8_t* tris, uint32_t triStride, const fyprt_mesh* meshes,
                               uint32_t meshCount, uint32_t kMaxDepth, SceneBVH& out) {
    out = SceneBVH();
    struct MeshOut { uint64_t forced = 0; std::vector<Node2> nodes; std::vector<Tri> tris; int32_t root = 0; uint32_t depth = 0; Box box; bool valid = false; };
    std::vector<MeshOut> mo(meshCount);
    // mesh bounds first: the TLAS only needs them, and its leaf depths set each BLAS's depth budget
    for (uint32_t m = 0; m < meshCount; ++m) {
        const fyprt_mesh& me = meshes[m];
        if (me.triangle_count == 0) continue;
        for (uint32_t i = 0; i < me.triangle_count; ++i) {
            const uint32_t* v = triIdx(tris, triStride, me.first_triangle + i);
            for (int k = 0; k < 3; ++k) mo[m].box.grow(verts[v[k]].position, verts[v[k]].position);
        }
        mo[m].valid = true;
    }
    std::vector<Prim> mp;
    for (uint32_t m = 0; m < meshCount; ++m) if (mo[m].valid) {
        Prim p; p.b = mo[m].box; p.id = m; for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.b.lo[a] + p.b.hi[a]); mp.push_back(p);
    }
    if (mp.empty()) return;
    std::vector<uint32_t> leafDepth(meshCount, 0);
    Builder tb; tb.maxLeaf = 1;
    tb.depthLimit = std::min(kMaxDepth - 2u, ceilLog2((uint32_t)mp.size()) + 6u);   // leave room for the BLASes
    tb.makeLeaf = [&](const Prim* p, uint32_t, uint32_t depth) -> int32_t { leafDepth[p[0].id] = depth; return INT32_MIN + (int32_t)p[0].id; };   // placeholder
    Box sceneBox;
    int32_t troot = tb.build(mp.data(), 0, (uint32_t)mp.size(), 0, sceneBox);
#pragma omp parallel for schedule(dynamic, 1)
    for (int m = 0; m < (int)meshCount; ++m) {
        const fyprt_mesh& me = meshes[m];
        if (me.triangle_count == 0) continue;
        std::vector<Prim> prims(me.triangle_count);
        for (uint32_t i = 0; i < me.triangle_count; ++i) {
            const uint32_t t = me.first_triangle + i; const uint32_t* v = triIdx(tris, triStride, t);
            Prim& p = prims[i]; p.id = t;
            for (int k = 0; k < 3; ++k) p.b.grow(verts[v[k]].position, verts[v[k]].position);
            for (int a = 0; a < 3; ++a) p.c[a] = 0.5f * (p.b.lo[a] + p.b.hi[a]);
        }
        MeshOut& o = mo[m];
        Builder b; b.maxLeaf = 4; b.depthLimit = kMaxDepth - leafDepth[m];
        if (const char* e = std::getenv("FYPRT_BVH_NODE_COST")) b.nodeCost = (float)std::atof(e);
        if (const char* e = std::getenv("FYPRT_BVH_BINS")) b.kBins = std::min(Builder::kMaxBins, std::max(2, std::atoi(e)));
        if (const char* e = std::getenv("FYPRT_BVH_SWEEP")) b.sweepBelow = (uint32_t)std::max(0, std::atoi(e));
        b.makeLeaf = [&](const Prim* p, uint32_t count, uint32_t) -> int32_t {
            const uint32_t first = (uint32_t)o.tris.size();
            for (uint32_t i = 0; i < count; ++i) {
                const uint32_t* v = triIdx(tris, triStride, p[i].id);
                const float *p0 = verts[v[0]].position, *p1 = verts[v[1]].position, *p2 = verts[v[2]].position;
                Tri t; std::memset(&t, 0, sizeof t);
                for (int a = 0; a < 3; ++a) { t.v0[a] = p0[a]; t.e1[a] = p1[a] - p0[a]; t.e2[a] = p2[a] - p0[a]; }
                t.tri = p[i].id;
                o.tris.push_back(t);
            }
            return ~(int32_t)((first << 2) | (count - 1));
        };
        Box bb;
        o.root = b.build(prims.data(), 0, (uint32_t)prims.size(), 0, bb);
        o.nodes.swap(b.nodes); o.depth = b.maxDepth; o.forced = b.forced;
    }
    // merge into one binary tree (TLAS nodes first, then each BLAS), relocating child references
    const uint32_t tlasNodes = (uint32_t)tb.nodes.size();
    std::vector<uint32_t> nodeOff(meshCount, 0), triOff(meshCount, 0);
    uint32_t no = tlasNodes, to = 0;
    for (uint32_t m = 0; m < meshCount; ++m) { nodeOff[m] = no; triOff[m] = to; no += (uint32_t)mo[m].nodes.size(); to += (uint32_t)mo[m].tris.size(); }
    auto relocate = [&](int32_t ref, uint32_t m) -> int32_t {
        if (ref >= 0) return ref + (int32_t)nodeOff[m];
        uint32_t code = (uint32_t)~ref; uint32_t first = (code >> 2) + triOff[m];
        return ~(int32_t)((first << 2) | (code & 3u));
    };
    auto resolveTlas = [&](int32_t ref) -> int32_t {
        if (ref >= 0) return ref;                                    // TLAS inner node (already global: TLAS is first)
        uint32_t m = (uint32_t)(ref - INT32_MIN);
        return relocate(mo[m].root, m);
    };
    std::vector<Node2> bin; bin.reserve(no); out.tris.reserve(to);
    for (Node2 n : tb.nodes) { n.child0 = resolveTlas(n.child0); n.child1 = resolveTlas(n.child1); bin.push_back(n); }
    for (uint32_t m = 0; m < meshCount; ++m) {
        for (Node2 n : mo[m].nodes) { n.child0 = relocate(n.child0, m); n.child1 = relocate(n.child1, m); bin.push_back(n); }
        out.tris.insert(out.tris.end(), mo[m].tris.begin(), mo[m].tris.end());
        if (mo[m].valid) out.maxDepth = std::max(out.maxDepth, leafDepth[m] + mo[m].depth);
    }
    const int32_t binRoot = resolveTlas(troot);
    out.binaryNodes = (uint32_t)bin.size();
    if (binRoot < 0) { out.rootRef = binRoot; return; }              // the whole scene is one leaf
    Collapser col(bin);
    col.computeHeights(binRoot);
    col.solve(binRoot);
    col.out.reserve(bin.size() / 2 + 1);
    uint32_t levels = 0;
    out.rootRef = col.emit(binRoot, levels);
    out.levels = levels;
    if (std::getenv("FYPRT_BVH_DEBUG")) { uint64_t f = tb.forced; uint32_t big = 0; for (uint32_t m = 0; m < meshCount; ++m) { f += mo[m].forced; big = std::max(big, meshes[m].triangle_count); } std::fprintf(stderr, "[bvh] forced median splits %llu, TLAS forced %llu, largest mesh %u tris\n", (unsigned long long)f, (unsigned long long)tb.forced, big); }
    if (std::getenv("FYPRT_BVH_DEBUG")) std::fprintf(stderr, "[bvh] binary nodes %zu height %u, wide nodes %zu, wide levels %u, tris %zu\n", bin.size(), col.h(binRoot), col.out.size(), levels, out.tris.size());
    if (std::getenv("FYPRT_BVH_DEBUG")) std::fprintf(stderr, "[bvh] collapse objective (sum of wide-node areas ~ expected node visits): width 2 %.4g, 4 %.4g, 6 %.4g, 8 %.4g\n",
                                                     collapseObjective<2>(bin, binRoot), collapseObjective<4>(bin, binRoot), collapseObjective<6>(bin, binRoot), collapseObjective<8>(bin, binRoot));
    out.nodes.swap(col.out);
    // Node order = decreasing box area (a priority-queue sweep from the root): the nodes a ray is most likely to visit form a prefix of
    // the array — the kernels keep that prefix in LDS (DevScene::topCount) — and hot nodes share cache lines further down.
    if (!std::getenv("FYPRT_BVH_PREORDER") && out.rootRef >= 0) {
        const size_t n = out.nodes.size();
        std::vector<int32_t> newIndex(n, -1); std::vector<int32_t> order; order.reserve(n);
        std::priority_queue<std::pair<float, int32_t>> pq;            // ties: the larger old index first (deterministic either way)
        pq.push({col.outArea[(size_t)out.rootRef], out.rootRef});
        while (!pq.empty()) {
            const int32_t old = pq.top().second; pq.pop();
            newIndex[(size_t)old] = (int32_t)order.size(); order.push_back(old);
            const Node& nd = out.nodes[(size_t)old];
            for (uint32_t i = 0; i < (nd.meta & 7u); ++i) if (nd.child[i] >= 0) pq.push({col.outArea[(size_t)nd.child[i]], nd.child[i]});
        }
        std::vector<Node> re(n);
        for (size_t k = 0; k < n; ++k) {
            Node nd = out.nodes[(size_t)order[k]];
            for (uint32_t i = 0; i < (nd.meta & 7u); ++i) if (nd.child[i] >= 0) nd.child[i] = newIndex[(size_t)nd.child[i]];
            re[k] = nd;
        }
        out.nodes.swap(re); out.rootRef = newIndex[(size_t)out.rootRef];
    }
}

void BuildSceneBVH(const fyprt_vertex* verts, const uint8_t* tris, uint32_t triStride, const fyprt_mesh* meshes,
                   uint32_t meshCount, SceneBVH& out) {
    BuildWithDepthBound(verts, tris, triStride, meshes, meshCount, kMaxDepthRelaxed, out);
    if (out.levels > kStackBudget || std::getenv("FYPRT_BVH_FORCE_SAFE_DEPTH")) BuildWithDepthBound(verts, tris, triStride, meshes, meshCount, kMaxDepthSafe, out);
}

}  // namespace rth
