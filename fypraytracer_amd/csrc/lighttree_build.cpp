// Host builder of the light trees in the reference's own shape, emitted in the flat node
// format of fyprt.h.  LIGHT_SOURCE_SAMPLING and NEE importance-sample lights by descending
// these trees (LightTree.cu:4-154), so the sampling distribution — and therefore per-pixel
// parity — depends on reproducing the reference's construction exactly, quirks included:
//   * LightTree::BuildHierarchyRecursively_SAOH (LightTree.cpp:21-293): 16-bin SAOH per axis,
//     bins start from AABB() = the origin box and ConeBounds() = a zero axis (LightTree.cuh:58-72),
//     so UnionCone's glm::rotate about a zero axis yields NaN cones for multi-emitter bins and
//     the cost comparison falls through to the median split (LightTree.cpp:226-252);
//   * the regulariser reads box[bestAxis] while bestAxis may still be -1 -> z (Vector3f.cuh:264-272);
//   * AABB::UnionAABB leaves the centroid of merged boxes at the origin (AABB.cuh:43-57), and the
//     device importance code reads that centroid (LightTree.cuh:101-105).
// Leaves: Mesh::CreateLightTreenodesFromEmmisiveMeshTriangles (Mesh.cpp:176-207); TLAS leaves:
// Scene::CreateLightTreeNodesFromBLASLightTrees (Scene.cpp:160-186).
// Host-only arithmetic, fp-contract off; acos / cos / sin are the deterministic binary64 routines of rt_hostmath.h (the device's own
// algorithms), NOT the C library's: the trees are the same bits on every host.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include "rt_host.h"
#include "rt_hostmath.h"

namespace rth {
namespace {

constexpr float kPi = 3.1415926535f;
struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross3(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 norm3(V3 a) { float inv = 1.0f / std::sqrt(dot3(a, a)); return a * inv; }
inline float vmaxf(float a, float b) { return (a < b) ? b : a; }   // glm::max

struct Cone { V3 axis{0, 0, 0}; float to = 0.0f, te = 0.0f; };
struct Bounds { V3 lo{0, 0, 0}, hi{0, 0, 0}, c{0, 0, 0}; };
inline Bounds unite(const Bounds& a, const Bounds& b) {            // centroid deliberately left at 0
    Bounds r;
    r.lo = {a.lo.x < b.lo.x ? a.lo.x : b.lo.x, a.lo.y < b.lo.y ? a.lo.y : b.lo.y, a.lo.z < b.lo.z ? a.lo.z : b.lo.z};
    r.hi = {a.hi.x > b.hi.x ? a.hi.x : b.hi.x, a.hi.y > b.hi.y ? a.hi.y : b.hi.y, a.hi.z > b.hi.z ? a.hi.z : b.hi.z};
    return r;
}
inline float area(const Bounds& b) { float dx = b.hi.x - b.lo.x, dy = b.hi.y - b.lo.y, dz = b.hi.z - b.lo.z; return 2.0f * (dx * dy + dy * dz + dz * dx); }
inline float axisOf(const V3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// glm::rotate(mat4(1), angle, axis) * vec4(v, 0)
V3 rotateAbout(V3 v, float angle, V3 axisIn) {
    float c, s; det_sincos(angle, s, c);
    const V3 a = norm3(axisIn), t = a * (1.0f - c);
    const V3 c0{c + t.x * a.x, t.x * a.y + s * a.z, t.x * a.z - s * a.y};
    const V3 c1{t.y * a.x - s * a.z, c + t.y * a.y, t.y * a.z + s * a.x};
    const V3 c2{t.z * a.x + s * a.y, t.z * a.y - s * a.x, c + t.z * a.z};
    const V3 zero{0.0f, 0.0f, 0.0f};
    return (c0 * v.x + c1 * v.y) + (c2 * v.z + zero * 0.0f);
}
Cone uniteCones(Cone a, Cone b) {                                   // ConeBounds::UnionCone
    if (b.to > a.to) std::swap(a, b);
    const float td = det_acos(dot3(a.axis, b.axis));
    const float te = std::fmax(a.te, b.te);
    if (std::fmin(td + b.to, kPi) <= a.to) return {a.axis, a.to, te};
    const float to = (a.to + td + b.to) * 0.5f;
    if (kPi <= to) return {a.axis, kPi, te};
    const float tr = to - a.to;
    return {norm3(rotateAbout(a.axis, tr, cross3(a.axis, b.axis))), to, te};
}
float orientMeasure(float to, float te) {                           // LightTree.cpp:318-329
    const float piHalf = 0.5f * kPi;
    const float tw = std::fmin(to + te, kPi);
    const float a = (2 * kPi) * (1 - det_cos(to));
    const float b = piHalf * (2 * tw * det_sin(to) - det_cos(to - 2 * tw) - (2 * to * det_sin(to)) + det_cos(to));
    return a + b;
}

struct Item { float energy; uint32_t num; Cone cone; Bounds box; V3 pos; uint32_t payload; };   // a light (leaf) awaiting placement

struct TreeOut { std::vector<fyprt_lighttree_node> nodes; };
struct NodeTmp { float energy; uint32_t num, left, right, leaf; Cone cone; Bounds box; };

uint32_t emit(TreeOut& o, const NodeTmp& n) {
    fyprt_lighttree_node f; std::memset(&f, 0, sizeof f);
    f.energy = n.energy; f.num_emitters = n.num; f.left = n.left; f.right_or_emitter = n.right; f.is_leaf = n.leaf;
    f.cone_axis[0] = n.cone.axis.x; f.cone_axis[1] = n.cone.axis.y; f.cone_axis[2] = n.cone.axis.z; f.theta_o = n.cone.to; f.theta_e = n.cone.te;
    f.box_lo[0] = n.box.lo.x; f.box_lo[1] = n.box.lo.y; f.box_lo[2] = n.box.lo.z; f.box_hi[0] = n.box.hi.x; f.box_hi[1] = n.box.hi.y; f.box_hi[2] = n.box.hi.z;
    f.box_centroid[0] = n.box.c.x; f.box_centroid[1] = n.box.c.y; f.box_centroid[2] = n.box.c.z;
    o.nodes.push_back(f);
    return (uint32_t)o.nodes.size() - 1;
}
NodeTmp fromFlat(const fyprt_lighttree_node& f) {
    NodeTmp n; n.energy = f.energy; n.num = f.num_emitters; n.left = f.left; n.right = f.right_or_emitter; n.leaf = f.is_leaf;
    n.cone.axis = {f.cone_axis[0], f.cone_axis[1], f.cone_axis[2]}; n.cone.to = f.theta_o; n.cone.te = f.theta_e;
    n.box.lo = {f.box_lo[0], f.box_lo[1], f.box_lo[2]}; n.box.hi = {f.box_hi[0], f.box_hi[1], f.box_hi[2]}; n.box.c = {f.box_centroid[0], f.box_centroid[1], f.box_centroid[2]};
    return n;
}

template <class It, class P> It partitionLikeLibstdcxx(It first, It last, P pred) {
    for (;;) {
        for (;;) { if (first == last) return first; if (pred(*first)) ++first; else break; }
        --last;
        for (;;) { if (first == last) return first; if (!pred(*last)) --last; else break; }
        std::iter_swap(first, last); ++first;
    }
}

uint32_t buildSaoh(TreeOut& o, Item* w, uint32_t first, uint32_t last) {
    const uint32_t count = last - first;
    if (count == 1) { NodeTmp n{w[first].energy, 1u, 0u, w[first].payload, 1u, w[first].cone, w[first].box}; return emit(o, n); }
    Bounds pb = w[first].box; Cone pc = w[first].cone; float pe = w[first].energy;
    for (uint32_t i = first + 1; i < last; ++i) pb = unite(pb, w[i].box);
    for (uint32_t i = first + 1; i < last; ++i) { pc = uniteCones(pc, w[i].cone); pe += w[i].energy; }
    float parentProb = area(pb) * orientMeasure(pc.to, pc.te) * pe;
    if (parentProb <= 0.0f) parentProb = 1e-12f;
    constexpr int B = 16;
    float bestCost = FLT_MAX; int bestAxis = -1, bestBin = -1;
    for (int axis = 0; axis < 3; ++axis) {
        float cmin = FLT_MAX, cmax = -FLT_MAX;
        for (uint32_t i = first; i < last; ++i) { const float v = axisOf(w[i].pos, axis); if (v < cmin) cmin = v; if (v > cmax) cmax = v; }
        if (cmin == cmax) continue;
        struct Bin { Bounds box; Cone cone; float energy = 0.0f; uint32_t num = 0; } bins[B];
        const float invRange = 1.0f / (cmax - cmin);
        for (uint32_t i = first; i < last; ++i) {
            int idx = (int)(((axisOf(w[i].pos, axis) - cmin) * invRange) * (float)(B - 1));
            idx = idx < 0 ? 0 : (idx > B - 1 ? B - 1 : idx);
            Bin& b = bins[idx];
            b.box = unite(b.box, w[i].box); b.cone = uniteCones(b.cone, w[i].cone); b.energy += w[i].energy; b.num += w[i].num;
        }
        Bounds sideBox[2][B - 1]; Cone sideCone[2][B - 1]; float sideE[2][B - 1]; uint32_t sideN[2][B - 1];
        for (int side = 0; side < 2; ++side) {
            bool any = false; Bounds cb; Cone cc; float ce = 0.0f; uint32_t cn = 0;
            for (int step = 0; step < B - 1; ++step) {
                const int bi = side == 0 ? step : (B - 1 - step), oi = side == 0 ? step : bi - 1;
                const Bin& b = bins[bi];
                if (b.num > 0) {
                    if (!any) { cb = b.box; cc = b.cone; ce = b.energy; cn = b.num; any = true; }
                    else { cb = unite(cb, b.box); cc = uniteCones(cc, b.cone); ce += b.energy; cn += b.num; }
                }
                sideBox[side][oi] = cb; sideCone[side][oi] = cc; sideE[side][oi] = ce; sideN[side][oi] = cn;
            }
        }
        for (int i = 0; i < B - 1; ++i) {
            if (sideN[0][i] == 0 || sideN[1][i] == 0) continue;
            const float pl = area(sideBox[0][i]) * orientMeasure(sideCone[0][i].to, sideCone[0][i].te) * sideE[0][i];
            const float pr = area(sideBox[1][i]) * orientMeasure(sideCone[1][i].to, sideCone[1][i].te) * sideE[1][i];
            float cost = (pl + pr) / parentProb;
            float lengthMax = pb.hi.x - pb.lo.x;
            lengthMax = vmaxf(lengthMax, pb.hi.y - pb.lo.y); lengthMax = vmaxf(lengthMax, pb.hi.z - pb.lo.z); lengthMax = vmaxf(lengthMax, 1e-12f);
            float ll = axisOf(sideBox[0][i].hi, bestAxis) - axisOf(sideBox[0][i].lo, bestAxis);
            float rl = axisOf(sideBox[1][i].hi, bestAxis) - axisOf(sideBox[1][i].lo, bestAxis);
            ll = vmaxf(ll, 1e-12f); rl = vmaxf(rl, 1e-12f);
            float kr = vmaxf(lengthMax / ll, lengthMax / rl);
            if (kr < 1.0f) kr = 1.0f;
            cost *= kr;
            if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = i; }
        }
    }
    uint32_t mid;
    if (bestAxis == -1) {
        mid = (first + last) / 2;
        std::nth_element(w + first, w + mid, w + last, [](const Item& a, const Item& b) { return a.pos.x < b.pos.x; });
    } else {
        float pmin = FLT_MAX, pmax = -FLT_MAX;
        for (uint32_t i = first; i < last; ++i) { const float v = axisOf(w[i].pos, bestAxis); if (v < pmin) pmin = v; if (v > pmax) pmax = v; }
        const float splitPos = pmin + (float)(bestBin + 1) * (pmax - pmin) / (float)B;
        Item* m = partitionLikeLibstdcxx(w + first, w + last, [&](const Item& n) { return axisOf(n.pos, bestAxis) < splitPos; });
        mid = (uint32_t)(m - w);
        if (mid == first || mid == last) mid = (first + last) / 2;
    }
    const uint32_t l = buildSaoh(o, w, first, mid), r = buildSaoh(o, w, mid, last);
    const NodeTmp L = fromFlat(o.nodes[l]), R = fromFlat(o.nodes[r]);
    NodeTmp p{L.energy + R.energy, L.num + R.num, l, r, 0u, uniteCones(L.cone, R.cone), unite(L.box, R.box)};
    return emit(o, p);
}

}  // namespace

// BLAS of one emissive mesh (Mesh.cpp:176-207 leaf producers + ConstructLightTree): nodes appended to `nodes`, returns the root index
// within them; `rootItem` = the TLAS leaf the mesh contributes (Scene.cpp:160-186)
static uint32_t buildMeshBlas(const fyprt_vertex* verts, const uint8_t* tris, uint32_t triStride, const fyprt_mesh& me, uint32_t meshIndex, float radiance,
                              std::vector<fyprt_lighttree_node>& nodes, Item& rootItem) {
    std::vector<Item> items(me.triangle_count);
    for (uint32_t i = 0; i < me.triangle_count; ++i) {
        const uint32_t t = me.first_triangle + i; const uint32_t* v = reinterpret_cast<const uint32_t*>(tris + (size_t)t * triStride);
        const fyprt_vertex &a = verts[v[0]], &b = verts[v[1]], &c = verts[v[2]];
        const V3 p0{a.position[0], a.position[1], a.position[2]}, p1{b.position[0], b.position[1], b.position[2]}, p2{c.position[0], c.position[1], c.position[2]};
        const V3 n0{a.normal[0], a.normal[1], a.normal[2]}, n1{b.normal[0], b.normal[1], b.normal[2]}, n2{c.normal[0], c.normal[1], c.normal[2]};
        Item& it = items[i];
        it.payload = t; it.num = 1; it.pos = ((p0 + p1) + p2) / 3.0f;
        auto mn = [](float x, float y) { return (y < x) ? y : x; }; auto mx = [](float x, float y) { return (x < y) ? y : x; };
        it.box.lo = {mn(mn(p0.x, p1.x), p2.x), mn(mn(p0.y, p1.y), p2.y), mn(mn(p0.z, p1.z), p2.z)};
        it.box.hi = {mx(mx(p0.x, p1.x), p2.x), mx(mx(p0.y, p1.y), p2.y), mx(mx(p0.z, p1.z), p2.z)};
        it.box.c = (it.box.lo + it.box.hi) * 0.5f;
        it.cone.te = kPi / 2.0f; it.cone.to = 0.0f; it.cone.axis = norm3(((n0 + n1) + n2) / 3.0f);
        const V3 cr = cross3(p1 - p0, p2 - p0);
        it.energy = ((0.5f * std::sqrt(dot3(cr, cr))) * radiance) * kPi;
    }
    TreeOut to;
    const uint32_t root = buildSaoh(to, items.data(), 0, (uint32_t)items.size());
    const NodeTmp rn = fromFlat(to.nodes[root]);
    rootItem.energy = rn.energy; rootItem.num = rn.num; rootItem.cone = rn.cone; rootItem.box = rn.box; rootItem.pos = rn.box.c; rootItem.payload = meshIndex;
    nodes.swap(to.nodes);
    return root;
}

// `touched` == nullptr: build everything.  Otherwise `out` holds the trees of the same topology and only the meshes with
// touched[m] != 0 get a new BLAS (a mesh's tree has 2n - 1 nodes whatever its geometry: replaced in place); the TLAS over the mesh
// trees' roots is always rebuilt (it is small).  The result equals a full build bit for bit: a mesh's BLAS depends on that mesh alone.
void BuildLightTrees(const fyprt_vertex* verts, const uint8_t* tris, uint32_t triStride, const fyprt_mesh* meshes,
                     uint32_t meshCount, const fyprt_material* mats, LightTrees& out, const uint8_t* touched) {
    const bool partial = touched != nullptr && out.first.size() == meshCount;
    if (!partial) { out = LightTrees(); out.first.assign(meshCount, 0); out.count.assign(meshCount, 0); out.root.assign(meshCount, ~0u); }
    std::vector<Item> tlasItems;
    for (uint32_t m = 0; m < meshCount; ++m) {
        const fyprt_mesh& me = meshes[m]; const fyprt_material& mat = mats[me.material_index];
        const V3 em{mat.emission_color[0] * mat.emission_power, mat.emission_color[1] * mat.emission_power, mat.emission_color[2] * mat.emission_power};
        if (!partial) out.first[m] = (uint32_t)out.blas.size();
        if (!(dot3(em, em) > 0.0f) || me.triangle_count == 0) continue;
        Item ti;
        if (partial && !touched[m]) {                              // untouched: the TLAS leaf comes from the BLAS root that is already there
            const NodeTmp rn = fromFlat(out.blas[out.first[m] + out.root[m]]);
            ti.energy = rn.energy; ti.num = rn.num; ti.cone = rn.cone; ti.box = rn.box; ti.pos = rn.box.c; ti.payload = m;
        } else {
            std::vector<fyprt_lighttree_node> nodes;
            const uint32_t root = buildMeshBlas(verts, tris, triStride, me, m, std::sqrt(dot3(em, em)), nodes, ti);
            if (partial) std::copy(nodes.begin(), nodes.end(), out.blas.begin() + out.first[m]);
            else out.blas.insert(out.blas.end(), nodes.begin(), nodes.end());
            out.count[m] = (uint32_t)nodes.size(); out.root[m] = root;
        }
        tlasItems.push_back(ti);
    }
    out.tlas.clear(); out.tlasRoot = ~0u;
    if (!tlasItems.empty()) {
        TreeOut to;
        // TLAS leaves keep the BLAS root's numEmitters (Scene.cpp:171-181 copies the node)
        out.tlasRoot = buildSaoh(to, tlasItems.data(), 0, (uint32_t)tlasItems.size());
        out.tlas.swap(to.nodes);
    }
}

}  // namespace rth
