// Device-side builder of the acceleration structure ("GPU LBVH builder + wide-node collapse"; the reference's own
// abandoned attempt is BVH.cu:1-279 / MortonCode.cuh:11-38).  Selected with tuning key 12 for the next fyprt_upload_scene.
//   k_lbvh_keys      63-bit Morton code of the triangle centroid (21 bits per axis over the scene box) as the sort key, the
//                    triangle index as the value (equal keys are told apart by their sorted position in the radix tree)
//   (radix sort)     hipcub::DeviceRadixSort::SortPairs
//   k_lbvh_radix     Karras 2012: one thread per internal node of the binary radix tree over the sorted keys — its key range and
//                    its two children
//   k_lbvh_boxes     boxes of the radix tree's nodes, bottom-up (second arrival at a node unites its children's boxes)
//   k_lbvh_collapse  one BFS level of the collapse into 4-wide nodes: a wide node starts from a binary node's two children and
//                    keeps replacing the inner child of largest surface area by its own two children; a subtree of <= 4
//                    triangles becomes a leaf (its triangles are consecutive in sorted order = in the leaf-triangle array)
//   k_lbvh_levels    reverse BFS: levels(node) = 1 + max(levels(inner children)) into the node's meta byte
// Boxes and their 8-bit quantisation are NOT computed here: the refit pass (rt_refit.h) does that for any topology.
// Tree quality is that of an LBVH (no SAH): traversal is slower than with the host builder; the build takes milliseconds.
//
// Key 12 = 2 replaces the radix tree by PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) over the same sorted
// order: the clusters — at first one per triangle, in Morton order — each look for the neighbour within `radius` positions whose
// union with them has the smallest surface area; mutual choices merge into a new binary node that takes the lower position; the
// list is compacted (order kept) and the round repeats until one cluster is left:
//   k_ploc_leaves    leaf boxes + the initial cluster list
//   k_ploc_nearest   one round's search, the window's boxes staged in LDS
//   k_ploc_merge     mutual pairs -> binary node (children, size, box); survivors flagged
//   (compaction)     hipcub::DeviceSelect::Flagged
// The result is a binary tree in the radix tree's node format (sizes instead of key ranges), so the same collapse runs on it.
#pragma once
#include "rt_device.h"

namespace rt {

RT_DEV unsigned long long expand_bits21(unsigned long long v) {   // 21 bits -> every third bit
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull; v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full; v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ void k_lbvh_keys(const float4* triPos, uint32_t nT, float3 lo, float3 invExt, unsigned long long* keys, uint32_t* vals) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nT) return;
    const float4* p = triPos + (size_t)t * 3;
    const float4 a = p[0], b = p[1], c = p[2];
    const float cx = ((a.x + b.x + c.x) * (1.0f / 3.0f) - lo.x) * invExt.x, cy = ((a.y + b.y + c.y) * (1.0f / 3.0f) - lo.y) * invExt.y,
                cz = ((a.z + b.z + c.z) * (1.0f / 3.0f) - lo.z) * invExt.z;
    auto q = [](float v) { v = v * 2097152.0f; v = v < 0.0f ? 0.0f : (v > 2097151.0f ? 2097151.0f : v); return (unsigned long long)(uint32_t)v; };   // NaN -> 0
    keys[t] = (expand_bits21(q(cx)) << 2) | (expand_bits21(q(cy)) << 1) | expand_bits21(q(cz));
    vals[t] = t;
}

// internal node i of the radix tree: range [first, last] of sorted leaves, children (bit 31 set = leaf index)
struct RadixNode { uint32_t first, last, left, right; };
constexpr uint32_t kRadixLeaf = 0x80000000u;

RT_DEV int lbvh_delta(const unsigned long long* keys, int n, int i, int j) {     // common prefix; equal keys are told apart by their position
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

__global__ void k_lbvh_radix(const unsigned long long* keys, int n, RadixNode* nodes, uint32_t* parentOfNode, uint32_t* parentOfLeaf) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n - 1) return;
    const int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0, t = l;
    do { t = (t + 1) >> 1; if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t; } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    RadixNode r; r.first = (uint32_t)lo; r.last = (uint32_t)hi;
    r.left = (lo == gamma) ? (kRadixLeaf | (uint32_t)gamma) : (uint32_t)gamma;
    r.right = (hi == gamma + 1) ? (kRadixLeaf | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    nodes[i] = r;
    if (r.left & kRadixLeaf) parentOfLeaf[r.left & ~kRadixLeaf] = (uint32_t)i; else parentOfNode[r.left] = (uint32_t)i;
    if (r.right & kRadixLeaf) parentOfLeaf[r.right & ~kRadixLeaf] = (uint32_t)i; else parentOfNode[r.right] = (uint32_t)i;
}

// boxes of the radix tree's internal nodes, bottom-up: every leaf walks towards the root, the second thread to arrive at a node
// (its sibling subtree is complete then) unites the two child boxes and goes on.  box[i] = {lo.xyz, hi.xyz}.
RT_DEV void lbvh_leaf_box(const float4* triPos, uint32_t tri, float* b) {
    const float4* p = triPos + (size_t)tri * 3;
    const float4 a = p[0], bb = p[1], c = p[2];
    b[0] = __builtin_fminf(a.x, __builtin_fminf(bb.x, c.x)); b[1] = __builtin_fminf(a.y, __builtin_fminf(bb.y, c.y)); b[2] = __builtin_fminf(a.z, __builtin_fminf(bb.z, c.z));
    b[3] = __builtin_fmaxf(a.x, __builtin_fmaxf(bb.x, c.x)); b[4] = __builtin_fmaxf(a.y, __builtin_fmaxf(bb.y, c.y)); b[5] = __builtin_fmaxf(a.z, __builtin_fmaxf(bb.z, c.z));
}
__global__ void k_lbvh_boxes(const RadixNode* rn, const uint32_t* parentOfNode, const uint32_t* parentOfLeaf, const uint32_t* vals, const float4* triPos,
                             uint32_t n, uint32_t* arrived, float* box) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t cur = parentOfLeaf[j];
    while (true) {
        __threadfence();
        if (atomicAdd(arrived + cur, 1u) == 0u) return;            // the sibling subtree is not done yet: its thread will do this node
        __threadfence();
        const RadixNode r = rn[cur];
        float a[6], b[6];
        if (r.left & kRadixLeaf) lbvh_leaf_box(triPos, vals[r.left & ~kRadixLeaf], a); else for (int k = 0; k < 6; ++k) a[k] = __hip_atomic_load(box + (size_t)r.left * 6 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r.right & kRadixLeaf) lbvh_leaf_box(triPos, vals[r.right & ~kRadixLeaf], b); else for (int k = 0; k < 6; ++k) b[k] = __hip_atomic_load(box + (size_t)r.right * 6 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < 3; ++k) { __hip_atomic_store(box + (size_t)cur * 6 + k, __builtin_fminf(a[k], b[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                      __hip_atomic_store(box + (size_t)cur * 6 + 3 + k, __builtin_fmaxf(a[3 + k], b[3 + k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (cur == 0u) return;
        cur = parentOfNode[cur];
    }
}

struct CollapseItem { uint32_t radix, wide, first; };      // binary node, its wide node, position of its first triangle in the leaf-triangle array
#ifndef RT_LBVH_LEAF
#define RT_LBVH_LEAF 4u
#endif
constexpr uint32_t kLbvhLeaf = RT_LBVH_LEAF;      // a radix subtree of at most this many triangles becomes one leaf

RT_DEV uint32_t radix_span(const RadixNode* rn, uint32_t ref) { return (ref & kRadixLeaf) ? 1u : rn[ref].last - rn[ref].first + 1u; }

// counters[0]: wide nodes allocated so far; counters[1]: items written to `out`
RT_DEV float lbvh_area(const float* box, uint32_t node) {
    const float* b = box + (size_t)node * 6;
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
// The triangles below a binary node occupy [first, first + span) of the leaf-triangle array, left subtree first — for the radix tree
// that is the sorted order, for a PLOC tree (whose clusters are not runs of the sorted order) it is what makes a leaf's triangles
// consecutive.  `vals` = triangle index per sorted position.
__global__ void k_lbvh_collapse(const RadixNode* rn, const float* box, const uint32_t* vals, const CollapseItem* in, uint32_t nIn, CollapseItem* out, uint32_t* counters,
                                float4* nodes, float4* leafTris) {
    static_assert(kLbvhLeaf <= 4u, "a leaf reference holds count - 1 in two bits");
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nIn) return;
    const CollapseItem it = in[k];
    uint32_t ch[4]; int cnt = 2;
    ch[0] = rn[it.radix].left; ch[1] = rn[it.radix].right;
    while (cnt < 4) {
        int best = -1; float bestArea = -1.0f;                    // the child of largest surface area among those of more than 4 triangles
        for (int i = 0; i < cnt; ++i) if (radix_span(rn, ch[i]) > kLbvhLeaf) { const float ar = lbvh_area(box, ch[i]); if (ar > bestArea) { bestArea = ar; best = i; } }
        if (best < 0) break;
        const RadixNode b = rn[ch[best]];
        for (int j = cnt; j > best + 1; --j) ch[j] = ch[j - 1];
        ch[best] = b.left; ch[best + 1] = b.right; ++cnt;
    }
    while (cnt < 4) {                                             // slots left over: split the would-be leaf of largest area into its two halves
        int best = -1; float bestArea = -1.0f;                    // (tighter boxes around fewer triangles at the price of nothing: the slots are tested anyway)
        for (int i = 0; i < cnt; ++i) if (!(ch[i] & kRadixLeaf) && radix_span(rn, ch[i]) <= kLbvhLeaf) { const float ar = lbvh_area(box, ch[i]); if (ar > bestArea) { bestArea = ar; best = i; } }
        if (best < 0) break;
        const RadixNode b = rn[ch[best]];
        for (int j = cnt; j > best + 1; --j) ch[j] = ch[j - 1];
        ch[best] = b.left; ch[best + 1] = b.right; ++cnt;
    }
    int32_t ref[4] = {(int32_t)0x80000000, (int32_t)0x80000000, (int32_t)0x80000000, (int32_t)0x80000000};
    uint32_t first = it.first;
    for (int i = 0; i < cnt; ++i) {
        const uint32_t sp = radix_span(rn, ch[i]);
        if (sp <= kLbvhLeaf) {
            ref[i] = ~(int32_t)((first << 2) | (sp - 1u));
            uint32_t stack[4], top = 0, pos = first;                  // the leaf's triangles, left to right
            stack[top++] = ch[i];
            while (top) {
                const uint32_t r = stack[--top];
                if (r & kRadixLeaf) { leafTris[(size_t)pos * 3 + 2] = make_float4(0.0f, __int_as_float((int)vals[r & ~kRadixLeaf]), 0.0f, 0.0f); ++pos; }   // k_refresh_leaf_tris fills in (v0, e1, e2)
                else { stack[top++] = rn[r].right; stack[top++] = rn[r].left; }
            }
        } else {
            const uint32_t w = atomicAdd(counters + 0, 1u);
            const uint32_t o = atomicAdd(counters + 1, 1u);
            out[o] = CollapseItem{ch[i], w, first};
            ref[i] = (int32_t)(w << 6);                                 // device form of an inner reference: the node's byte offset (rt_host.h)
        }
        first += sp;
    }
    float4* n = nodes + (size_t)it.wide * 4;
    n[0] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float((int)((uint32_t)cnt << 24)));      // boxes + exponents: the refit pass
    n[1] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), __int_as_float(ref[2]), __int_as_float(ref[3]));
    n[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); n[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}


// ------------------------------------------------------------------------------------------------ PLOC (key 12 = 2)
// Boxes: 6 floats per binary node; internal node k at slot k (what lbvh_area reads), leaf j (sorted position) at slot nLeaves + j.
RT_DEV uint32_t ploc_slot(uint32_t ref, uint32_t nLeaves) { return (ref & kRadixLeaf) ? nLeaves + (ref & ~kRadixLeaf) : ref; }

__global__ void k_ploc_leaves(const uint32_t* vals, const float4* triPos, uint32_t n, float* box, uint32_t* clusters) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float b[6];
    lbvh_leaf_box(triPos, vals[j], b);
    for (int k = 0; k < 6; ++k) box[(size_t)(n + j) * 6 + k] = b[k];
    clusters[j] = kRadixLeaf | j;
}

constexpr int kPlocBlock = 256, kPlocMaxRadius = 32;
// nearest[i] = position of the cluster within `radius` whose union with cluster i has the smallest area; ties: the partner i ^ 1
// first, then the lower position (with that, equal boxes pair up instead of forming a chain of one-sided choices).  force != 0:
// everybody takes i ^ 1 — the host's way out of a round that merged almost nothing.
__global__ __launch_bounds__(kPlocBlock) void k_ploc_nearest(const uint32_t* clusters, uint32_t n, const float* box, uint32_t nLeaves, int radius, int force, uint32_t* nearest) {
    __shared__ float s_box[(kPlocBlock + 2 * kPlocMaxRadius) * 6];
    const int base = (int)(blockIdx.x * kPlocBlock) - radius;
    for (int t = (int)threadIdx.x; t < kPlocBlock + 2 * radius; t += kPlocBlock) {
        const int g = base + t;
        if (g >= 0 && g < (int)n) { const float* b = box + (size_t)ploc_slot(clusters[g], nLeaves) * 6; for (int k = 0; k < 6; ++k) s_box[t * 6 + k] = b[k]; }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * kPlocBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t partner = ((i ^ 1u) < n) ? (i ^ 1u) : i - 1u;         // n >= 2
    const float* me = s_box + ((int)threadIdx.x + radius) * 6;
    auto cost = [&](int g) {
        const float* o = s_box + (g - base) * 6;
        const float dx = __builtin_fmaxf(me[3], o[3]) - __builtin_fminf(me[0], o[0]), dy = __builtin_fmaxf(me[4], o[4]) - __builtin_fminf(me[1], o[1]),
                    dz = __builtin_fmaxf(me[5], o[5]) - __builtin_fminf(me[2], o[2]);
        return dx * dy + dy * dz + dz * dx;
    };
    uint32_t best = partner;
    if (!force) {
        float bestCost = cost((int)partner);
        const int lo = (int)i - radius < 0 ? 0 : (int)i - radius, hi = (int)i + radius >= (int)n ? (int)n - 1 : (int)i + radius;
        for (int g = lo; g <= hi; ++g) {
            if (g == (int)i) continue;
            const float cg = cost(g);
            if (cg < bestCost) { bestCost = cg; best = (uint32_t)g; }      // NaN boxes never win: `best` stays a valid position
        }
    }
    nearest[i] = best;
}

// mutual choices merge: the lower position keeps the new node, the upper one drops out of the list
__global__ void k_ploc_merge(const uint32_t* clusters, const uint32_t* nearest, uint32_t n, uint32_t nLeaves, RadixNode* rn, float* box, uint32_t* nodeCount,
                             uint32_t* next, uint8_t* keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = nearest[i];
    const uint32_t mine = clusters[i];
    if (nearest[j] != i) { next[i] = mine; keep[i] = 1; return; }
    if (i > j) { next[i] = mine; keep[i] = 0; return; }
    const uint32_t theirs = clusters[j];
    const uint32_t k = atomicAdd(nodeCount, 1u);
    const float* a = box + (size_t)ploc_slot(mine, nLeaves) * 6; const float* b = box + (size_t)ploc_slot(theirs, nLeaves) * 6;
    float* o = box + (size_t)k * 6;
    for (int c = 0; c < 3; ++c) { o[c] = __builtin_fminf(a[c], b[c]); o[3 + c] = __builtin_fmaxf(a[3 + c], b[3 + c]); }
    RadixNode r; r.left = mine; r.right = theirs; r.first = 0u;
    r.last = radix_span(rn, mine) + radix_span(rn, theirs) - 1u;          // span = last - first + 1 (the collapse reads nothing else of the range)
    rn[k] = r;
    next[i] = k; keep[i] = 1;
}

// nodes [first, first + count) of one BFS level, deepest level first
__global__ void k_lbvh_levels(float4* nodes, uint32_t first, uint32_t count) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    float4* n = nodes + (size_t)(first + k) * 4;
    const float4 q0 = n[0], q1 = n[1];
    const uint32_t ex = (uint32_t)__float_as_int(q0.w), cnt = (ex >> 24) & 7u;
    const int32_t child[4] = {__float_as_int(q1.x), __float_as_int(q1.y), __float_as_int(q1.z), __float_as_int(q1.w)};
    uint32_t below = 0;
    for (uint32_t i = 0; i < cnt; ++i) if (child[i] >= 0) {
        const uint32_t cl = ((uint32_t)__float_as_int(nodes[(size_t)(child[i] >> 6) * 4].w)) >> 27;
        below = cl > below ? cl : below;
    }
    const uint32_t levels = 1u + below;
    n[0].w = __int_as_float((int)((ex & 0x07FFFFFFu) | ((levels > 31u ? 31u : levels) << 27)));
}

}  // namespace rt
