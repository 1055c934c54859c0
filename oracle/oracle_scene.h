// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_vec.h header).
//
// Scene-side data and the host producers the kernel inputs depend on, restated from
//   Vertex.h:5-10, Triangle.cuh:7-60, Material.cuh:7-21 / Material.cu:5-18, AABB.cuh:9-77,
//   Texture.cu:103-139, Scene.cpp:9-92 (triangle AABBs, mesh AABBs), Scene.cpp:209-221,
//   BVH.cuh:27-69 + BVH.cpp:65-108,146-309 (binned SAH, 1 triangle per leaf),
//   LightTree.cuh:28-73 + LightTree.cpp:4-340 (binned SAOH), ConeBounds.cuh:14-87,
//   Mesh.cpp:176-207 and Scene.cpp:160-186 (light-tree leaves).
// Deterministic readings fixed here (the reference is racy / implementation-defined):
//   * the three SAH axes are evaluated serially 0,1,2 and ties keep the lowest axis
//     (the reference runs them under `#pragma omp parallel for` + critical, BVH.cpp:168-265);
//   * std::partition is restated with the libstdc++ bidirectional algorithm;
//   * LightTree regulariser reads box[bestAxis] with bestAxis == -1 -> z (Vector3f.cuh:264-272).
#pragma once
#include <algorithm>
#include <cfloat>
#include <vector>
#include "oracle_math.h"

namespace orc {

struct Vertex { vec3 position, normal; vec2 uv; };                        // Vertex.h:5-10 (32 B)
struct TriIdx { uint32_t v0, v1, v2; int32_t materialIndex; };           // Triangle.cuh:9-10
struct Material {                                                         // Material.cuh:7-16 (44 B)
    uint32_t isUseAlbedoMap; vec3 albedo; uint32_t albedoMapIndex;
    float roughness, metallic; vec3 emissionColor; float emissionPower;
    vec3  GetEmission() const { return emissionColor * emissionPower; }              // Material.cu:5-8
    float GetEmissionRadiance() const { return length(emissionColor * emissionPower); } // :10-13
};
struct MeshRange { uint32_t firstTriangle, triangleCount; int32_t materialIndex; }; // Mesh.h:26-31
struct Texture { const uint32_t* pixels; uint32_t width, height; };

struct AABB {                                                             // AABB.cuh:9-77
    vec3 lo{0, 0, 0}, hi{0, 0, 0}, centroid{0, 0, 0};
    static AABB Union(const AABB& a, const AABB& b) {                     // :43-57 (centroid NOT updated)
        AABB c;
        c.lo = v3(a.lo.x < b.lo.x ? a.lo.x : b.lo.x, a.lo.y < b.lo.y ? a.lo.y : b.lo.y, a.lo.z < b.lo.z ? a.lo.z : b.lo.z);
        c.hi = v3(a.hi.x > b.hi.x ? a.hi.x : b.hi.x, a.hi.y > b.hi.y ? a.hi.y : b.hi.y, a.hi.z > b.hi.z ? a.hi.z : b.hi.z);
        return c;
    }
    static vec3 FindCentroid(AABB& a) { a.centroid = (a.lo + a.hi) * 0.5f; return a.centroid; }  // :37-41
    float SurfaceArea() const {                                            // :70-76
        float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

// ---- Triangle helpers (Triangle.cuh:14-59)
static inline vec3 TriCentroid(vec3 p0, vec3 p1, vec3 p2) { return (p0 + p1 + p2) / 3.0f; }   // "GetBarycentricCoords"
static inline vec3 TriRandomPoint(vec3 p0, vec3 p1, vec3 p2, uint32_t& seed) {
    float r1 = randomFloat(seed), r2 = randomFloat(seed);
    float s = sqrtf(r1);
    float u = 1.0f - s, v = (1.0f - r2) * s, w = r2 * s;
    return u * p0 + v * p1 + w * p2;
}
static inline vec3 TriNormal(vec3 n0, vec3 n1, vec3 n2) { return normalize((n0 + n1 + n2) / 3.0f); }
static inline float TriArea(vec3 p0, vec3 p1, vec3 p2) { return 0.5f * length(cross(p1 - p0, p2 - p0)); }

// ---- Texture::SampleBilinear (Texture.cu:103-139)
static inline uint32_t SampleBilinear(const Texture& t, float u, float v) {
    u = (u < 0.0f) ? 0.0f : (u > 1.0f ? 1.0f : u);
    v = (v < 0.0f) ? 0.0f : (v > 1.0f ? 1.0f : v);
    float x = u * (float)(t.width - 1), y = v * (float)(t.height - 1);
    int x0 = (int)x, y0 = (int)y;
    int x1 = (x0 + 1 < (int)t.width) ? x0 + 1 : x0, y1 = (y0 + 1 < (int)t.height) ? y0 + 1 : y0;
    float tx = x - (float)x0, ty = y - (float)y0;
    vec4 c00 = UnpackABGR(t.pixels[y0 * t.width + x0]), c10 = UnpackABGR(t.pixels[y0 * t.width + x1]);
    vec4 c01 = UnpackABGR(t.pixels[y1 * t.width + x0]), c11 = UnpackABGR(t.pixels[y1 * t.width + x1]);
    vec4 cx0 = c00 * (1.0f - tx) + c10 * tx, cx1 = c01 * (1.0f - tx) + c11 * tx;
    return ConvertToRGBA(cx0 * (1.0f - ty) + cx1 * ty);
}

// =========================================================================== BVH (reference format)
struct BVHNode { AABB box; uint32_t objectIndex = ~0u, child1 = ~0u, child2 = ~0u; bool isLeaf = false; };  // BVH.cuh:27-60
struct BVH { std::vector<BVHNode> nodes; uint32_t rootIndex = ~0u; };

template <class It, class Pred> static It partition_bidir(It first, It last, Pred pred) {
    while (true) {
        while (true) { if (first == last) return first; else if (pred(*first)) ++first; else break; }
        --last;
        while (true) { if (first == last) return first; else if (!pred(*last)) --last; else break; }
        std::iter_swap(first, last); ++first;
    }
}

static uint32_t BuildSAH(std::vector<BVHNode>& out, BVHNode* work, size_t first, size_t last) {   // BVH.cpp:146-309
    const size_t count = last - first;
    if (count == 1) { BVHNode n; n.box = work[first].box; n.objectIndex = work[first].objectIndex; n.isLeaf = true; out.push_back(n); return (uint32_t)out.size() - 1; }
    AABB bounds = work[first].box;                                                   // RangeBounds :100-108
    for (size_t i = first + 1; i < last; ++i) bounds = AABB::Union(bounds, work[i].box);
    const int numBins = 16;
    float bestCost = FLT_MAX; int bestAxis = -1, bestSplitBin = -1;
    for (int axis = 0; axis < 3; ++axis) {
        float cmin = FLT_MAX, cmax = -FLT_MAX;
        for (size_t i = first; i < last; ++i) { float c = comp(AABB::FindCentroid(work[i].box), axis); cmin = (cmin < c) ? cmin : c; cmax = (cmax > c) ? cmax : c; }
        if (cmin == cmax) continue;
        struct Bin { AABB box; size_t count = 0; } bins[numBins];                     // zero box incl. origin (:192-201 quirk)
        for (size_t i = first; i < last; ++i) {
            float c = comp(AABB::FindCentroid(work[i].box), axis);
            int b = (int)(((c - cmin) / (cmax - cmin)) * (float)(numBins - 1));
            bins[b].count++; bins[b].box = AABB::Union(bins[b].box, work[i].box);
        }
        AABB leftBoxes[numBins - 1], rightBoxes[numBins - 1]; size_t leftCounts[numBins - 1], rightCounts[numBins - 1];
        { AABB cur; cur.lo = v3(FLT_MAX); cur.hi = v3(-FLT_MAX); size_t cnt = 0;
          for (int i = 0; i < numBins - 1; ++i) { cur = AABB::Union(cur, bins[i].box); cnt += bins[i].count; leftBoxes[i] = cur; leftCounts[i] = cnt; } }
        { AABB cur; cur.lo = v3(FLT_MAX); cur.hi = v3(-FLT_MAX); size_t cnt = 0;
          for (int i = numBins - 1; i > 0; --i) { cur = AABB::Union(cur, bins[i].box); cnt += bins[i].count; rightBoxes[i - 1] = cur; rightCounts[i - 1] = cnt; } }
        float parentArea = bounds.SurfaceArea();
        for (int i = 0; i < numBins - 1; ++i) {
            if (leftCounts[i] == 0 || rightCounts[i] == 0) continue;
            float cost = 1.0f + ((float)leftCounts[i] * leftBoxes[i].SurfaceArea() + (float)rightCounts[i] * rightBoxes[i].SurfaceArea()) / parentArea;
            if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplitBin = i; }
        }
    }
    size_t mid;
    if (bestAxis == -1) {                                                             // :269-284
        mid = (first + last) / 2;
        std::nth_element(work + first, work + mid, work + last, [](const BVHNode& a, const BVHNode& b) {
            AABB A = a.box, B = b.box; return AABB::FindCentroid(A).x < AABB::FindCentroid(B).x; });
    } else {
        float cmin = comp(bounds.lo, bestAxis), cmax = comp(bounds.hi, bestAxis);     // box bounds, not centroid bounds (:287-289)
        float splitPos = cmin + (float)(bestSplitBin + 1) * (cmax - cmin) / (float)numBins;
        BVHNode* m = partition_bidir(work + first, work + last, [&](const BVHNode& n) { AABB B = n.box; return comp(AABB::FindCentroid(B), bestAxis) < splitPos; });
        mid = (size_t)(m - work);
        if (mid == first || mid == last) mid = (first + last) / 2;
    }
    uint32_t l = BuildSAH(out, work, first, mid), r = BuildSAH(out, work, mid, last);
    BVHNode p; p.box = AABB::Union(out[l].box, out[r].box); p.child1 = l; p.child2 = r; p.isLeaf = false;
    out.push_back(p); return (uint32_t)out.size() - 1;
}
static inline void ConstructBVH_SAH(BVH& bvh, std::vector<BVHNode>& objects) {         // BVH.cpp:65-81
    bvh.nodes.clear(); bvh.rootIndex = ~0u;
    if (objects.empty()) return;
    bvh.nodes.reserve(2 * objects.size() - 1);
    bvh.rootIndex = BuildSAH(bvh.nodes, objects.data(), 0, objects.size());
}

// =========================================================================== light tree (reference format)
struct ConeBounds { vec3 axis{0, 0, 0}; float theta_o = 0.0f, theta_e = 0.0f; };       // ConeBounds.cuh:9-13

// glm::rotate(mat4(1), angle, axis) applied to (v, 0)   (ConeBounds.cuh:40-43); host-only, libm trig.
static inline vec3 RotateAbout(vec3 v, float angle, vec3 axisIn) {
    const float c = t_cos(angle), s = t_sin(angle);
    vec3 axis = normalize(axisIn);
    vec3 temp = axis * (1.0f - c);
    // glm::rotate builds Rotate[col][row]; Result = m * Rotate with m = identity
    vec3 c0 = v3(c + temp.x * axis.x, temp.x * axis.y + s * axis.z, temp.x * axis.z - s * axis.y);
    vec3 c1 = v3(temp.y * axis.x - s * axis.z, c + temp.y * axis.y, temp.y * axis.z + s * axis.x);
    vec3 c2 = v3(temp.z * axis.x + s * axis.y, temp.z * axis.y - s * axis.x, c + temp.z * axis.z);
    // R * vec4(v,0): (c0*x + c1*y) + (c2*z + c3*0)
    vec3 a = c0 * v.x + c1 * v.y, b = c2 * v.z + v3(0.0f) * 0.0f;
    return a + b;
}
static inline ConeBounds UnionCone(ConeBounds a, ConeBounds b) {                       // ConeBounds.cuh:14-45
    if (b.theta_o > a.theta_o) std::swap(a, b);
    float theta_d = t_acos(dot(a.axis, b.axis));
    float theta_e = fmaxf(a.theta_e, b.theta_e);
    if (fminf(theta_d + b.theta_o, kPi) <= a.theta_o) return {a.axis, a.theta_o, theta_e};
    float theta_o = (a.theta_o + theta_d + b.theta_o) * 0.5f;
    if (kPi <= theta_o) return {a.axis, kPi, theta_e};
    float theta_r = theta_o - a.theta_o;
    vec3 rotAxis = cross(a.axis, b.axis);
    vec3 axis = normalize(RotateAbout(a.axis, theta_r, rotAxis));
    return {axis, theta_o, theta_e};
}

struct LTNode {                                                                        // LightTree.cuh:28-49
    float energy = 0.0f; uint32_t numEmitters = 0, offset = 0; ConeBounds bounds_o; AABB bounds_w; vec3 position{0, 0, 0};
    bool isLeaf = false; uint32_t emitterIndex = ~0u;
};
struct LightTree { std::vector<LTNode> nodes; uint32_t rootIndex = ~0u; };

static inline float OrientMeasure(float theta_o, float theta_e) {                      // LightTree.cpp:318-329
    const float piHalf = 0.5f * kPi;
    float theta_w = fminf(theta_o + theta_e, kPi);
    float a = (2 * kPi) * (1 - t_cos(theta_o));
    float b = piHalf * (2 * theta_w * t_sin(theta_o) - t_cos(theta_o - 2 * theta_w) - (2 * theta_o * t_sin(theta_o)) + t_cos(theta_o));
    return a + b;
}

static uint32_t BuildSAOH(std::vector<LTNode>& out, LTNode* work, uint32_t first, uint32_t last) {   // LightTree.cpp:21-293
    const uint32_t count = last - first;
    if (count == 1) {
        LTNode n; n.energy = work[first].energy; n.numEmitters = 1; n.offset = 0; n.bounds_o = work[first].bounds_o;
        n.bounds_w = work[first].bounds_w; n.position = work[first].position; n.isLeaf = true; n.emitterIndex = work[first].emitterIndex;
        out.push_back(n); return (uint32_t)out.size() - 1;
    }
    AABB parentBounds = work[first].bounds_w;
    for (uint32_t i = first + 1; i < last; ++i) parentBounds = AABB::Union(parentBounds, work[i].bounds_w);
    ConeBounds parentCone = work[first].bounds_o; float parentEnergy = work[first].energy;
    for (uint32_t i = first + 1; i < last; ++i) { parentCone = UnionCone(parentCone, work[i].bounds_o); parentEnergy += work[i].energy; }
    float parentProb = parentBounds.SurfaceArea() * OrientMeasure(parentCone.theta_o, parentCone.theta_e) * parentEnergy;
    if (parentProb <= 0.0f) parentProb = 1e-12f;
    const int numBins = 16;
    float bestCost = FLT_MAX; int bestAxis = -1, bestSplitBin = -1;
    for (int axis = 0; axis < 3; ++axis) {
        float cmin = FLT_MAX, cmax = -FLT_MAX;
        for (uint32_t i = first; i < last; ++i) { float v = comp(work[i].position, axis); if (v < cmin) cmin = v; if (v > cmax) cmax = v; }
        if (cmin == cmax) continue;
        struct Bin { AABB bounds_w; ConeBounds bounds_o; float energy = 0.0f; uint32_t numEmitters = 0; } bins[numBins];
        const float invRange = 1.0f / (cmax - cmin);
        for (uint32_t i = first; i < last; ++i) {
            float v = comp(work[i].position, axis);
            int idx = iclamp((int)(((v - cmin) * invRange) * (float)(numBins - 1)), 0, numBins - 1);
            Bin& b = bins[idx];                                                       // Bin::AddEmitter, LightTree.cuh:66-72
            b.bounds_w = AABB::Union(b.bounds_w, work[i].bounds_w); b.bounds_o = UnionCone(b.bounds_o, work[i].bounds_o);
            b.energy += work[i].energy; b.numEmitters += work[i].numEmitters;
        }
        AABB lB[numBins - 1], rB[numBins - 1]; ConeBounds lC[numBins - 1], rC[numBins - 1];
        float lE[numBins - 1], rE[numBins - 1]; uint32_t lN[numBins - 1], rN[numBins - 1];
        auto sweep = [&](bool leftSide) {
            bool any = false; AABB curA; ConeBounds curC; float curE = 0.0f; uint32_t curN = 0;
            for (int s = 0; s < numBins - 1; ++s) {
                int bi = leftSide ? s : (numBins - 1 - s);          // bins 0..14 left-to-right, 15..1 right-to-left
                int oi = leftSide ? s : (bi - 1);
                if (!any && bins[bi].numEmitters > 0) { curA = bins[bi].bounds_w; curC = bins[bi].bounds_o; curE = bins[bi].energy; curN = bins[bi].numEmitters; any = true; }
                else if (any && bins[bi].numEmitters > 0) { curA = AABB::Union(curA, bins[bi].bounds_w); curC = UnionCone(curC, bins[bi].bounds_o); curE += bins[bi].energy; curN += bins[bi].numEmitters; }
                if (leftSide) { lB[oi] = curA; lC[oi] = curC; lE[oi] = curE; lN[oi] = curN; }
                else          { rB[oi] = curA; rC[oi] = curC; rE[oi] = curE; rN[oi] = curN; }
            }
        };
        sweep(true); sweep(false);
        for (int i = 0; i < numBins - 1; ++i) {
            if (lN[i] == 0 || rN[i] == 0) continue;
            float P_left = lB[i].SurfaceArea() * OrientMeasure(lC[i].theta_o, lC[i].theta_e) * lE[i];
            float P_right = rB[i].SurfaceArea() * OrientMeasure(rC[i].theta_o, rC[i].theta_e) * rE[i];
            float cost = (P_left + P_right) / parentProb;
            float lengthMax = parentBounds.hi.x - parentBounds.lo.x;
            lengthMax = gmax(lengthMax, parentBounds.hi.y - parentBounds.lo.y);
            lengthMax = gmax(lengthMax, parentBounds.hi.z - parentBounds.lo.z);
            lengthMax = gmax(lengthMax, 1e-12f);
            float leftLen = comp(lB[i].hi, bestAxis) - comp(lB[i].lo, bestAxis);      // bestAxis may still be -1 -> z (:202-203)
            float rightLen = comp(rB[i].hi, bestAxis) - comp(rB[i].lo, bestAxis);
            leftLen = gmax(leftLen, 1e-12f); rightLen = gmax(rightLen, 1e-12f);
            float kr = gmax(lengthMax / leftLen, lengthMax / rightLen);
            if (kr < 1.0f) kr = 1.0f;
            cost *= kr;
            if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestSplitBin = i; }
        }
    }
    uint32_t mid;
    if (bestAxis == -1) {
        mid = (first + last) / 2;
        std::nth_element(work + first, work + mid, work + last, [](const LTNode& a, const LTNode& b) { return a.position.x < b.position.x; });
    } else {
        float pmin = FLT_MAX, pmax = -FLT_MAX;
        for (uint32_t i = first; i < last; ++i) { float v = comp(work[i].position, bestAxis); if (v < pmin) pmin = v; if (v > pmax) pmax = v; }
        float splitPos = pmin + (float)(bestSplitBin + 1) * (pmax - pmin) / (float)numBins;
        LTNode* m = partition_bidir(work + first, work + last, [&](const LTNode& n) { return comp(n.position, bestAxis) < splitPos; });
        mid = (uint32_t)(m - work);
        if (mid == first || mid == last) mid = (first + last) / 2;
    }
    uint32_t l = BuildSAOH(out, work, first, mid), r = BuildSAOH(out, work, mid, last);
    LTNode p; p.isLeaf = false; p.offset = l; p.emitterIndex = r;
    p.bounds_w = AABB::Union(out[l].bounds_w, out[r].bounds_w);                       // centroid stays (0,0,0): AABB.cuh:43-57
    p.bounds_o = UnionCone(out[l].bounds_o, out[r].bounds_o);
    p.energy = out[l].energy + out[r].energy; p.numEmitters = out[l].numEmitters + out[r].numEmitters;
    out.push_back(p); return (uint32_t)out.size() - 1;
}
static inline void ConstructLightTree(LightTree& t, std::vector<LTNode>& objects) {     // LightTree.cpp:4-19
    t.nodes.clear(); t.rootIndex = ~0u;
    if (objects.empty()) return;
    t.nodes.reserve(2 * objects.size() - 1);
    t.rootIndex = BuildSAOH(t.nodes, objects.data(), 0, (uint32_t)objects.size());
}

// =========================================================================== scene container
struct Scene {
    std::vector<Vertex> worldVertices; std::vector<TriIdx> triangles; std::vector<AABB> triBoxes;
    std::vector<Material> materials; std::vector<MeshRange> meshes; std::vector<Texture> textures;
    std::vector<std::vector<uint32_t>> texturePixels;
    std::vector<uint32_t> emissiveTriangles;
    std::vector<BVH> blas; BVH tlas;
    std::vector<LightTree> lightBlas; LightTree lightTlas;

    void Build() {
        // triangle AABBs (Scene.cpp:61-75)
        triBoxes.resize(triangles.size());
        for (size_t i = 0; i < triangles.size(); ++i) {
            vec3 p0 = worldVertices[triangles[i].v0].position, p1 = worldVertices[triangles[i].v1].position, p2 = worldVertices[triangles[i].v2].position;
            triBoxes[i].lo = vmin(vmin(p0, p1), p2); triBoxes[i].hi = vmax(vmax(p0, p1), p2);
            AABB::FindCentroid(triBoxes[i]);
        }
        // emissive list (Scene.cpp:209-221)
        emissiveTriangles.clear();
        for (uint32_t i = 0; i < triangles.size(); ++i)
            if (length2(materials[triangles[i].materialIndex].GetEmission()) > 0.0f) emissiveTriangles.push_back(i);
        // per-mesh BLAS (Mesh.cpp:148-174 + ConstructBVH_SAH) and mesh AABB (Scene.cpp:79-87)
        blas.assign(meshes.size(), BVH()); lightBlas.assign(meshes.size(), LightTree());
        std::vector<BVHNode> tlasLeaves;
        for (size_t m = 0; m < meshes.size(); ++m) {
            const MeshRange& mr = meshes[m];
            std::vector<BVHNode> leaves(mr.triangleCount);
            for (uint32_t i = 0; i < mr.triangleCount; ++i) { leaves[i].objectIndex = mr.firstTriangle + i; leaves[i].box = triBoxes[mr.firstTriangle + i]; leaves[i].isLeaf = true; }
            ConstructBVH_SAH(blas[m], leaves);
            BVHNode ml; ml.objectIndex = (uint32_t)m; ml.isLeaf = true;
            if (mr.triangleCount > 0) {
                AABB mb = triBoxes[mr.firstTriangle + mr.triangleCount - 1];
                for (uint32_t i = 0; i < mr.triangleCount; ++i) mb = AABB::Union(mb, triBoxes[mr.firstTriangle + i]);
                AABB::FindCentroid(mb); ml.box = mb;
            }
            tlasLeaves.push_back(ml);
            // per-mesh light tree (Mesh.cpp:176-207)
            if (length2(materials[mr.materialIndex].GetEmission()) > 0.0f) {
                std::vector<LTNode> ll;
                for (uint32_t i = 0; i < mr.triangleCount; ++i) {
                    uint32_t ti = mr.firstTriangle + i; const TriIdx& t = triangles[ti];
                    const Vertex &a = worldVertices[t.v0], &b = worldVertices[t.v1], &c = worldVertices[t.v2];
                    LTNode n; n.emitterIndex = ti; n.position = TriCentroid(a.position, b.position, c.position); n.bounds_w = triBoxes[ti];
                    n.bounds_o.theta_e = kPi / 2.0f; n.bounds_o.theta_o = 0.0f; n.bounds_o.axis = TriNormal(a.normal, b.normal, c.normal);
                    n.energy = TriArea(a.position, b.position, c.position) * materials[mr.materialIndex].GetEmissionRadiance() * kPi;
                    n.numEmitters = 1; n.isLeaf = true; ll.push_back(n);
                }
                ConstructLightTree(lightBlas[m], ll);
            }
        }
        ConstructBVH_SAH(tlas, tlasLeaves);
        // light-tree TLAS leaves (Scene.cpp:160-186)
        std::vector<LTNode> tl;
        for (uint32_t m = 0; m < meshes.size(); ++m)
            if (!lightBlas[m].nodes.empty()) {
                LTNode n = lightBlas[m].nodes[lightBlas[m].rootIndex];
                n.emitterIndex = m; n.offset = 0; n.position = n.bounds_w.centroid; n.isLeaf = true;
                tl.push_back(n);
            }
        ConstructLightTree(lightTlas, tl);
    }
};

}  // namespace orc
