// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_vec.h header).
//
// CPU restatement of the reference hot path, function for function:
//   Renderer.cu:460-561  TraceRay (two-level LIFO traversal, Möller–Trumbore)
//   BVH.cuh:124-165      IntersectRayAABB
//   Renderer.cu:2389-2429 ClosestHit / Miss
//   Renderer.cu:565-1284 PerPixel_{BruteForce,Uniform,Cosine,GGX,BRDF}
//   Renderer.cu:1287-1626 PerPixel_LightSourceSampling / PerPixel_NextEventEstimation
//   Renderer.cu:1628-2041 PerPixel_ReSTIR_DI_Part1 / Part2
//   Renderer.cu:2043-2387 PerPixel_ReSTIR_GI_Part1 / Part2
//   Renderer.cu:2431-2900 kernel epilogues + the vec4(0) sentinel protocol
//   LightTree.cuh:91-117, LightTree.cu:4-276, ConeBounds.cuh:47-87
//   ReSTIR_DI_Reservoir.cu:3-50, ReSTIR_GI_Reservoir.cu:5-68
//   Camera.cpp:136-153   per-pixel ray directions
//
// PARITY PIN: the reference ships no tests, golden vectors or runnable build for this path
//  — "parity unpinned" by the reference; the closed-form known answers
// of SURVEY.md §8c are checked in tests/test_oracle_kat.py and everything else is pinned
// by fixtures generated from this restatement (tests/golden/).
//
// Deterministic readings where the reference is racy / undefined:
//   R1  Miss() zero-fills worldPosition and sets objectIndex = -1 (uninitialised in Ray.h:13-22).
//   R2  The octahedral normal G-buffer is double-buffered: Part 1 reads the previous
//       frame's normals and writes the current frame's for every pixel; Part 2 reads the
//       current frame's (the reference has one buffer written and read concurrently).
//   R3  DI temporal history clamp (Renderer.cu:1776-1779) is applied to a local copy of the
//       previous reservoir, not written back through the reference.
//   R4  `albedoMapIndex <= textureCount - 1` with textureCount == 0 wraps to "always true"
//       in the reference and then reads textures[] out of bounds; guarded as "no texture".
//   R5  float -> int conversions that are UB for NaN / out-of-range (reprojection floor)
//       are defined as NaN -> 0 and saturate.
//   R7  Temporal history is used only where it exists and means something: (i) a tile-split render (rows [y0, y1) + halo)
//       holds history for the rows it rendered LAST frame only — a reprojection landing outside them reads as "no history"
//       (the whole frame when not split: no change to the reference); (ii) a DI history light index that is not in the
//       current emissive list (scene replaced) reads as "no history" (the reference reads out of bounds).
#pragma once
#include "oracle_scene.h"

namespace orc {

struct Ray { vec3 origin, direction; };
struct Payload { float hitDistance; vec3 worldPosition, worldNormal; float u, v; int32_t objectIndex; };  // Ray.h:13-22, 40 B
struct DIReservoir { uint32_t indexEmissive; float weightEmissive, emissivePDF, weightSum; uint32_t M; }; // 20 B
struct GISample { vec3 visiblePoint; vec2 visibleNormal; vec3 samplePoint; vec2 sampleNormal; vec3 Lo; uint32_t randSeed; float samplePDF; };
struct GIReservoir { GISample sample; float weightSample; uint32_t M; float weightSum; };                  // 72 B
static_assert(sizeof(Payload) == 40 && sizeof(DIReservoir) == 20 && sizeof(GIReservoir) == 72, "layout");

struct Settings {                                                          // RenderingSettings.h:5-22 (52 B)
    uint8_t toAccumulate, _p0[3]; int32_t lightBounces, sampleCount; vec3 skyColor; int32_t technique, lightCandidateCount;
    uint32_t randSeed; uint8_t useTemporalReuse, useSpatialReuse, _p1[2]; int32_t temporalHistoryLimit, spatialNeighborNum, spatialNeighborRadius;
};
static_assert(sizeof(Settings) == 52, "layout");
enum { BRUTE_FORCE, UNIFORM_SAMPLING, COSINE_WEIGHTED_SAMPLING, GGX_SAMPLING, BRDF_SAMPLING, LIGHT_SOURCE_SAMPLING, NEE, RESTIR_DI, RESTIR_GI };

struct Camera { mat4 projection, view, prevProjection, prevView, inverseProjection, inverseView; vec3 position; uint32_t width, height; };

struct Counters { uint64_t rays = 0, boxTests = 0, triTests = 0, hits = 0, nodeVisits = 0; };

// ---- reservoirs
static inline bool DI_Update(DIReservoir& r, uint32_t cand, float weight, uint32_t count, float pdf, uint32_t& seed) { // DI.cu:3-36
    r.weightSum += weight; r.M += count;
    if (randomFloat(seed) < weight / r.weightSum) { r.indexEmissive = cand; r.emissivePDF = pdf; return true; }
    return false;
}
static inline void DI_Reset(DIReservoir& r) { r.indexEmissive = 0; r.weightEmissive = 0.0f; r.weightSum = 0.0f; r.M = 0; r.emissivePDF = 0.0f; }
static inline void GI_ResetSample(GISample& s) { s.visiblePoint = v3(0.0f); s.visibleNormal = {0, 0}; s.samplePoint = v3(0.0f); s.sampleNormal = {0, 0}; s.Lo = v3(0.0f); s.randSeed = 0; s.samplePDF = 0.0f; }
static inline void GI_Reset(GIReservoir& r) { GI_ResetSample(r.sample); r.weightSample = 0.0f; r.M = 0; r.weightSum = 0.0f; }
static inline bool GI_Update(GIReservoir& r, const GISample& s, float w, uint32_t count, float pdf, uint32_t& seed) {   // GI.cu:5-34
    r.weightSum += w; r.M += count;
    if (randomFloat(seed) < w / r.weightSum) { r.sample = s; r.sample.samplePDF = pdf; return true; }
    return false;
}
static inline bool GI_Merge(GIReservoir& r, const GIReservoir& o, float pdf, uint32_t& seed) {                         // GI.cu:36-43
    uint32_t prev = r.M;
    bool upd = GI_Update(r, o.sample, pdf * o.weightSum * (float)o.M, 1, pdf, seed);
    r.M = prev + o.M;
    return upd;
}
static inline bool GI_Valid(const GIReservoir& r) { return r.M > 0 && length2(r.sample.Lo) > 0.0f; }                  // GI.cu:65-68

// ---- tracer interface (reference traversal below; the product's own traversal order is
//      restated in oracle_product_trace.h for the instrumented byte counts)
struct Tracer {
    virtual ~Tracer() {}
    virtual Payload Trace(const Ray& ray, Counters& c) const = 0;
    // Shadow query: the callers only use (objectIndex == target && hitDistance >= 0) | hitDistance < 0 | other
    // (R.cu:2016-2031, :1386-1394, :1503-1504).  The reference answers it with a full closest-hit TraceRay — the
    // default here; the product's early-terminating variant is restated in ProductTracer.
    virtual Payload TraceTo(const Ray& ray, uint32_t targetTri, Counters& c) const { (void)targetTri; return Trace(ray, c); }
    // GI visibility query (R.cu:2356-2366): visible iff |t_closest - dist| <= tol.  The reference finds the closest hit —
    // the default here; the product's interval-cut / early-out variant is restated in ProductTracer.
    virtual bool TraceVisible(const Ray& ray, float dist, float tol, Counters& c) const {
        Payload p = Trace(ray, c);
        return fabsf(p.hitDistance - dist) <= tol;
    }
};

static inline bool IntersectRayAABB(const Ray& ray, const AABB& box) {          // BVH.cuh:124-165
    float tMin = 0.0f, tMax = FLT_MAX;
    for (int i = 0; i < 3; ++i) {
        float o = comp(ray.origin, i), d = comp(ray.direction, i);
        if (d == 0.0f) { if (o < comp(box.lo, i) || o > comp(box.hi, i)) return false; continue; }
        float invD = 1.0f / d;
        float t0 = (comp(box.lo, i) - o) * invD, t1 = (comp(box.hi, i) - o) * invD;
        if (invD < 0.0f) { float t = t0; t0 = t1; t1 = t; }
        tMin = fmaxf(tMin, t0); tMax = fminf(tMax, t1);
        if (tMax < tMin) return false;
    }
    return true;
}
static inline Payload Miss() { Payload p; p.hitDistance = -1.0f; p.worldPosition = v3(0.0f); p.worldNormal = v3(0.0f); p.u = 0.0f; p.v = 0.0f; p.objectIndex = -1; return p; } // :2423-2429 + R1
static inline Payload ClosestHit(const Scene& sc, const Ray& ray, float t, int tri, float u, float v) {   // :2389-2421
    Payload p; p.hitDistance = t; p.objectIndex = tri;
    const TriIdx& T = sc.triangles[tri];
    p.worldPosition = ray.origin + ray.direction * t;
    float w = 1.0f - u - v;
    const Vertex &a = sc.worldVertices[T.v0], &b = sc.worldVertices[T.v1], &c = sc.worldVertices[T.v2];
    p.worldNormal = normalize(a.normal * w + b.normal * u + c.normal * v);
    vec2 uv = a.uv * w + b.uv * u + c.uv * v;
    p.u = uv.x; p.v = uv.y;
    return p;
}
// Möller–Trumbore exactly as Renderer.cu:513-537 (no parallel-ray epsilon, no culling).
static inline bool IntersectTri(const Ray& ray, vec3 v0, vec3 v1, vec3 v2, float closest, float& tOut, float& uOut, float& vOut) {
    vec3 e1 = v1 - v0, e2 = v2 - v0;
    vec3 h = cross(ray.direction, e2);
    float a = dot(e1, h);
    float f = 1.0f / a;
    vec3 s = ray.origin - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    vec3 q = cross(s, e1);
    float v = f * dot(ray.direction, q);
    if (v < 0.0f || (u + v) > 1.0f) return false;
    float t = f * dot(e2, q);
    if (t > 0.0001f && t < closest) { tOut = t; uOut = u; vOut = v; return true; }
    return false;
}

struct ReferenceTracer : Tracer {                                                // Renderer.cu:460-561
    const Scene& sc; explicit ReferenceTracer(const Scene& s) : sc(s) {}
    Payload Trace(const Ray& ray, Counters& c) const override {
        c.rays++;
        if (sc.triangles.empty()) return Miss();
        float closest = FLT_MAX; int closestTri = -1; float cu = 0.0f, cv = 0.0f;
        int tlasStack[256]; int tTop = 0; static thread_local std::vector<int> blasStackStore(1024); int* blasStack = blasStackStore.data();
        if (sc.tlas.nodes.empty()) return Miss();
        tlasStack[tTop++] = (int)sc.tlas.rootIndex;
        while (tTop > 0) {
            const BVHNode& node = sc.tlas.nodes[tlasStack[--tTop]];
            c.boxTests++;
            if (!IntersectRayAABB(ray, node.box)) continue;
            if (node.isLeaf) {
                const BVH& blas = sc.blas[node.objectIndex];
                if (blas.nodes.empty()) continue;
                int bTop = 0; blasStack[bTop++] = (int)blas.rootIndex;
                while (bTop > 0) {
                    const BVHNode& bn = blas.nodes[blasStack[--bTop]];
                    c.boxTests++;
                    if (!IntersectRayAABB(ray, bn.box)) continue;
                    if (bn.isLeaf) {
                        const TriIdx& T = sc.triangles[bn.objectIndex];
                        c.triTests++;
                        float t, u, v;
                        if (IntersectTri(ray, sc.worldVertices[T.v0].position, sc.worldVertices[T.v1].position, sc.worldVertices[T.v2].position, closest, t, u, v)) {
                            closest = t; closestTri = (int)bn.objectIndex; cu = u; cv = v;
                        }
                    } else {
                        if (bn.child1 != ~0u && bTop < 1024) blasStack[bTop++] = (int)bn.child1;
                        if (bn.child2 != ~0u && bTop < 1024) blasStack[bTop++] = (int)bn.child2;
                    }
                }
            } else {
                if (node.child1 != ~0u && tTop < 256) tlasStack[tTop++] = (int)node.child1;
                if (node.child2 != ~0u && tTop < 256) tlasStack[tTop++] = (int)node.child2;
            }
        }
        if (closestTri < 0) return Miss();
        c.hits++;
        return ClosestHit(sc, ray, closest, closestTri, cu, cv);
    }
};

// ---- light tree traversal (LightTree.cuh:91-117, ConeBounds.cuh:47-87, LightTree.cu)
struct ShadingPoint { vec3 position, normal; };
struct SampledLight { uint32_t emitterIndex; float pmf; };

static inline float ConeThetaToAABB(const AABB& aabb, vec3 p) {                 // FindConeThatEnvelopsAABBFromPoint -> theta_o
    vec3 axis = normalize(aabb.centroid - p);
    float maxTheta = 0.0f;
    for (int i = 0; i < 8; ++i) {
        vec3 corner = v3((i & 4) ? aabb.hi.x : aabb.lo.x, (i & 2) ? aabb.hi.y : aabb.lo.y, (i & 1) ? aabb.hi.z : aabb.lo.z);
        vec3 dir = normalize(corner - p);
        float cosT = gclamp(dot(axis, dir), -1.0f, 1.0f);
        maxTheta = fmaxf(maxTheta, t_acos(cosT));
    }
    return maxTheta;
}
static inline float ClusterImportance(const ShadingPoint& sp, const LTNode& cl) {   // LightTree.cuh:91-117
    float theta_u = ConeThetaToAABB(cl.bounds_w, sp.position);
    vec3 dir = sp.position - cl.bounds_w.centroid;
    float d2 = fmaxf(dot(dir, dir), 1e-12f);
    dir = normalize(dir);
    float dotVal = gclamp(dot(cl.bounds_o.axis, dir), -1.0f, 1.0f);
    float theta = t_acos(dotVal);
    float angleTerm = gclamp(theta - cl.bounds_o.theta_o - theta_u, 0.0f, cl.bounds_o.theta_e);
    return (cl.energy * t_cos(angleTerm)) / d2;
}
// one descent step shared by PickLight_TLAS / PickLight_BLAS (LightTree.cu:25-77, :101-153)
static inline uint32_t DescendStep(const LightTree& t, uint32_t nodeIdx, const ShadingPoint& sp, float& rnd, float& pmfAcc) {
    const LTNode& node = t.nodes[nodeIdx];
    uint32_t l = node.offset, r = node.emitterIndex;
    float Il = ClusterImportance(sp, t.nodes[l]), Ir = ClusterImportance(sp, t.nodes[r]);
    float sum = Il + Ir;
    if (!(sum > 0.0f) || (Il + Ir) <= 0.0f) { sum = 1.0f; Il = 0.5f; }
    float p_left = gclamp(Il / sum, 1e-6f, 1.0f - 1e-6f);
    if (rnd < p_left) { pmfAcc *= p_left; rnd = rnd / p_left; return l; }
    float p_right = 1.0f - p_left; pmfAcc *= p_right; rnd = (rnd - p_left) / p_right; return r;
}
static inline SampledLight PickLight_BLAS(const LightTree& t, const ShadingPoint& sp, float rnd, float currentPMF) {  // LightTree.cu:4-78
    SampledLight out{~0u, 0.0f};
    if (t.nodes.empty() || t.rootIndex == ~0u) return out;
    uint32_t idx = t.rootIndex; float pmfAcc = 1.0f;
    rnd = gclamp(rnd, 0.0f, 0.9999999f);
    while (!t.nodes[idx].isLeaf) idx = DescendStep(t, idx, sp, rnd, pmfAcc);
    out.emitterIndex = t.nodes[idx].emitterIndex; out.pmf = currentPMF * pmfAcc;
    return out;
}
static inline SampledLight PickLight_TLAS(const Scene& sc, const ShadingPoint& sp, uint32_t& seed) {                  // LightTree.cu:80-154
    SampledLight out{~0u, 0.0f};
    float rnd = randomFloat(seed);
    const LightTree& t = sc.lightTlas;
    if (t.nodes.empty() || t.rootIndex == ~0u) return out;
    uint32_t idx = t.rootIndex; float pmfAcc = 1.0f;
    rnd = gclamp(rnd, 0.0f, 0.9999999f);
    while (!t.nodes[idx].isLeaf) idx = DescendStep(t, idx, sp, rnd, pmfAcc);
    return PickLight_BLAS(sc.lightBlas[t.nodes[idx].emitterIndex], sp, rnd, pmfAcc);
}
static inline float ComputeDirectEmitterPMF(const Scene& sc, const ShadingPoint& sp, uint32_t emitterIndex) {          // LightTree.cu:156-276
    const LightTree& t = sc.lightTlas;
    if (t.nodes.empty() || t.rootIndex == ~0u) return 0.0f;
    uint32_t tlasLeaf = ~0u;
    for (uint32_t i = 0; i < t.nodes.size() && tlasLeaf == ~0u; ++i) {
        if (!t.nodes[i].isLeaf) continue;
        const LightTree& b = sc.lightBlas[t.nodes[i].emitterIndex];
        for (uint32_t j = 0; j < b.nodes.size(); ++j)
            if (b.nodes[j].isLeaf && b.nodes[j].emitterIndex == emitterIndex) { tlasLeaf = i; break; }
    }
    if (tlasLeaf == ~0u) return 0.0f;
    float pmfAcc = 1.0f; uint32_t idx = t.rootIndex;
    auto step = [&](const LightTree& tr, uint32_t target) {                     // index-range heuristic (:227, :263), bug-for-bug
        uint32_t l = tr.nodes[idx].offset, r = tr.nodes[idx].emitterIndex;
        float Il = ClusterImportance(sp, tr.nodes[l]), Ir = ClusterImportance(sp, tr.nodes[r]);
        float sum = Il + Ir;
        if (!(sum > 0.0f)) { Il = 0.5f; sum = 1.0f; }
        float p_left = Il / sum;
        if (l <= target && target <= l + (tr.nodes[l].numEmitters - 1)) { pmfAcc *= p_left; idx = l; }
        else { pmfAcc *= (1.0f - p_left); idx = r; }
    };
    while (!t.nodes[idx].isLeaf) step(t, tlasLeaf);
    const LightTree& b = sc.lightBlas[t.nodes[idx].emitterIndex];
    if (b.nodes.empty()) return 0.0f;
    idx = b.rootIndex;
    while (!b.nodes[idx].isLeaf) step(b, emitterIndex);
    return pmfAcc;
}

// =========================================================================== per-frame state + render
struct Frame {
    uint32_t W = 0, H = 0, frameIndex = 1;
    uint32_t histDI[2] = {0, 0}, histGI[2] = {0, 0};                             // R7 (i): rows holding DI / GI history
    std::vector<vec4> accum; std::vector<uint32_t> image;
    std::vector<Payload> payload; std::vector<float> depth; std::vector<vec2> normalPrev, normalCur;
    std::vector<DIReservoir> di, diPrev; std::vector<GIReservoir> gi, giPrev;
    void Resize(uint32_t w, uint32_t h) {                                        // Renderer.cpp:5-41, Renderer.cu:286-419
        W = w; H = h; size_t n = (size_t)w * h; frameIndex = 1;
        accum.assign(n, vec4{0, 0, 0, 0}); image.assign(n, 0u);
        Payload z; std::memset(&z, 0, sizeof z); payload.assign(n, z); depth.assign(n, 0.0f);
        normalPrev.assign(n, vec2{0, 0}); normalCur.assign(n, vec2{0, 0});
        DIReservoir dz; std::memset(&dz, 0, sizeof dz); di.assign(n, dz); diPrev.assign(n, dz);
        GIReservoir gz; std::memset(&gz, 0, sizeof gz); gi.assign(n, gz); giPrev.assign(n, gz);
        histDI[0] = histGI[0] = 0; histDI[1] = histGI[1] = h;
    }
};

struct Renderer {
    const Scene& sc; const Tracer& tracer; Camera cam; Settings st; Frame fr;
    bool skipDeadShadowRays = false;   // the product's tuning key 18 (only meaningful beside the ProductTracer: ray counts must be the product's)
    Renderer(const Scene& s, const Tracer& t) : sc(s), tracer(t) {}

    vec3 RayDirection(uint32_t x, uint32_t y) const {                            // Camera.cpp:136-153
        vec2 coord{(float)x / (float)cam.width, (float)y / (float)cam.height};
        coord = coord * 2.0f + (-1.0f);
        vec4 target = cam.inverseProjection * vec4{coord.x, coord.y, 1.0f, 1.0f};
        vec3 d = normalize(xyz(target) / target.w);
        return xyz(cam.inverseView * v4(d, 0.0f));
    }
    vec3 Albedo(const Material& m, float u, float v) const {                     // e.g. Renderer.cu:607-621 (+R4)
        if (m.isUseAlbedoMap && !sc.textures.empty() && m.albedoMapIndex <= (uint32_t)sc.textures.size() - 1) {
            vec4 c = UnpackABGR(SampleBilinear(sc.textures[m.albedoMapIndex], u, v));
            return v3(c.x, c.y, c.z);
        }
        return m.albedo;
    }
    const Material& MatOf(int tri) const { return sc.materials[sc.triangles[tri].materialIndex]; }

    // ---- techniques 0..4 (Renderer.cu:565-1284); one body, the sampler is the only difference
    vec4 PerPixel_Path(uint32_t x, uint32_t y, int tech, Counters& c) const {
        const uint8_t maxBounces = (uint8_t)st.lightBounces, sampleCount = (uint8_t)st.sampleCount;
        uint32_t seed = x + y * fr.W; seed *= fr.frameIndex;
        vec3 radiance = v3(0.0f);
        Ray primary{cam.position, RayDirection(x, y)};
        Payload pp = tracer.Trace(primary, c);
        if (pp.hitDistance < 0.0f) return v4(st.skyColor, 1.0f);
        const Material& hm = MatOf(pp.objectIndex);
        if (length(hm.GetEmission()) > 0.0f) return v4(hm.GetEmission(), 1.0f);
        const int nSamples = (tech == BRUTE_FORCE) ? 1 : (int)sampleCount;
        auto sampleDir = [&](const Payload& hit, vec3 V, const Material& m, vec3 albedo, float rough, float& pdf) -> vec3 {
            vec3 d;
            switch (tech) {
                case BRUTE_FORCE: case UNIFORM_SAMPLING: d = UniformSampleHemisphere(hit.worldNormal, seed); pdf = UniformHemispherePDF(); return d;
                case COSINE_WEIGHTED_SAMPLING: d = CosineSampleHemisphere(hit.worldNormal, seed); pdf = -1.0f; return d;   // pdf from cosTheta below
                case GGX_SAMPLING: return GGXSampleHemisphere(hit.worldNormal, V, rough, seed, pdf);
                default: return BRDFSampleHemisphere(hit.worldNormal, V, albedo, m.metallic, m.roughness, seed, pdf);
            }
        };
        for (int s = 0; s < nSamples; ++s) {
            if (tech != BRUTE_FORCE) seed += (uint32_t)((s + 1) * 27);
            vec3 T = v3(1.0f);
            Payload hit = pp;
            vec3 albedo = Albedo(hm, hit.u, hit.v);
            float pdf;
            vec3 dir = sampleDir(pp, -primary.direction, hm, albedo, hm.roughness, pdf);
            vec3 brdf = CalculateBRDF(pp.worldNormal, -primary.direction, dir, albedo, hm.metallic, hm.roughness);
            float cosT = gmax(dot(dir, pp.worldNormal), 0.0f);
            if (tech == COSINE_WEIGHTED_SAMPLING) pdf = CosineHemispherePDF(cosT);
            T *= brdf * cosT / pdf;
            Ray ray{pp.worldPosition + pp.worldNormal * 1e-12f, dir};
            for (int b = 0; b < (int)maxBounces; ++b) {
                seed += (uint32_t)((tech == BRUTE_FORCE ? 0 : s) + 31 * b);
                hit = tracer.Trace(ray, c);
                if (hit.hitDistance < 0.0f) { radiance += T * st.skyColor; break; }
                const Material& m = MatOf(hit.objectIndex);
                vec3 em = m.GetEmission();
                if (length(em) > 0.0f) { radiance += T * em; break; }
                vec3 alb = Albedo(m, hit.u, hit.v);
                float bpdf;
                vec3 bdir = sampleDir(hit, -ray.direction, m, alb, hm.roughness /* primary roughness: :1091-1092 */, bpdf);
                vec3 bbrdf = CalculateBRDF(hit.worldNormal, -ray.direction, bdir, alb, m.metallic, m.roughness);
                float bcos = gmax(dot(bdir, hit.worldNormal), 0.0f);
                if (tech == COSINE_WEIGHTED_SAMPLING) bpdf = CosineHemispherePDF(bcos);
                T *= bbrdf * bcos / bpdf;
                ray.origin = hit.worldPosition + hit.worldNormal * 1e-12f; ray.direction = bdir;
            }
        }
        if (tech != BRUTE_FORCE) radiance /= (float)sampleCount;
        return v4(radiance, 1.0f);
    }

    // ---- LIGHT_SOURCE_SAMPLING (Renderer.cu:1287-1408)
    vec4 PerPixel_LightSource(uint32_t x, uint32_t y, Counters& c) const {
        const uint8_t sampleCount = (uint8_t)st.sampleCount;
        uint32_t seed = x + y * fr.W; seed *= fr.frameIndex;
        vec3 radiance = v3(0.0f);
        Ray primary{cam.position, RayDirection(x, y)};
        Payload pp = tracer.Trace(primary, c);
        if (pp.hitDistance < 0.0f) return v4(st.skyColor, 1.0f);
        const Material& hm = MatOf(pp.objectIndex);
        if (length(hm.GetEmission()) > 0.0f) return v4(hm.GetEmission(), 1.0f);
        for (int s = 0; s < (int)sampleCount; ++s) {
            seed += (uint32_t)((s + 1) * 27);
            vec3 T = v3(1.0f);
            ShadingPoint sp{pp.worldPosition, pp.worldNormal};
            SampledLight sl = PickLight_TLAS(sc, sp, seed);
            const TriIdx& lt = sc.triangles[sl.emitterIndex];
            const Vertex &a = sc.worldVertices[lt.v0], &b = sc.worldVertices[lt.v1], &cc = sc.worldVertices[lt.v2];
            vec3 ep = TriRandomPoint(a.position, b.position, cc.position, seed);
            vec3 dir = ep - pp.worldPosition;
            float dist = distance(ep, pp.worldPosition);
            dir = dir / dist;
            vec3 albedo = Albedo(hm, pp.u, pp.v);
            vec3 brdf = CalculateBRDF(pp.worldNormal, -primary.direction, dir, albedo, hm.metallic, hm.roughness);
            float cx = gmax(dot(dir, pp.worldNormal), 0.0f);
            float cy = gmax(dot(-dir, TriNormal(a.normal, b.normal, cc.normal)), 0.0f);
            float triAreaPDF = 1.0f / TriArea(a.position, b.position, cc.position);
            float totalPDF = sl.pmf * triAreaPDF * (dist * dist);
            T *= brdf * cx * cy / totalPDF;
            Ray ray{pp.worldPosition + pp.worldNormal * 1e-12f, dir};
            Payload hit = tracer.TraceTo(ray, sl.emitterIndex, c);
            if (hit.hitDistance < 0.0f) { radiance += T * st.skyColor; continue; }
            if ((uint32_t)hit.objectIndex != sl.emitterIndex) continue;
            const Material& m = sc.materials[lt.materialIndex];
            if (m.GetEmissionRadiance() > 0.0f) radiance += T * m.GetEmission();
        }
        radiance /= (float)sampleCount;
        return v4(radiance, 1.0f);
    }

    // ---- NEE (Renderer.cu:1411-1626)
    vec4 PerPixel_NEE(uint32_t x, uint32_t y, Counters& c) const {
        const uint8_t maxBounces = (uint8_t)st.lightBounces, sampleCount = (uint8_t)st.sampleCount;
        uint32_t seed = (x + y * fr.W) * fr.frameIndex;
        vec3 radiance = v3(0.0f);
        Ray ray{cam.position, RayDirection(x, y)};
        Payload payload = tracer.Trace(ray, c);
        if (payload.hitDistance < 0.0f) return v4(st.skyColor, 1.0f);
        const Material& hm = MatOf(payload.objectIndex);
        if (length(hm.GetEmission()) > 0.0f) return v4(hm.GetEmission(), 1.0f);
        for (int s = 0; s < (int)sampleCount; ++s) {
            seed += (uint32_t)((s + 1) * 31);
            vec3 T = v3(1.0f); Ray pathRay = ray; Payload hit = payload;
            float pdfBRDF = 1.0f, pdfDirect = 1.0f;
            for (int bounce = 0; bounce < (int)maxBounces; ++bounce) {
                const Material& mat = MatOf(hit.objectIndex);
                vec3 albedo = Albedo(mat, hit.u, hit.v);
                ShadingPoint sp{hit.worldPosition, hit.worldNormal};
                SampledLight sl = PickLight_TLAS(sc, sp, seed);
                const TriIdx& lt = sc.triangles[sl.emitterIndex];
                vec3 p0 = sc.worldVertices[lt.v0].position, p1 = sc.worldVertices[lt.v1].position, p2 = sc.worldVertices[lt.v2].position;
                vec3 n0 = sc.worldVertices[lt.v0].normal, n1 = sc.worldVertices[lt.v1].normal, n2 = sc.worldVertices[lt.v2].normal;
                vec3 lp = TriRandomPoint(p0, p1, p2, seed);
                vec3 ld = lp - hit.worldPosition;
                float dist = length(ld);
                ld /= dist;
                Ray shadow{hit.worldPosition + hit.worldNormal * 1e-12f, ld};
                // product twin only (skipDeadShadowRays): the product prepares the direct term before the ray and does not trace a ray whose term
                // is exactly zero — the sum below would not change; the pixel is the reference's, the ray counters are the product's
                bool dead = false;
                if (skipDeadShadowRays) {
                    vec3 ln = TriNormal(n0, n1, n2);
                    vec3 brdf = CalculateBRDF(hit.worldNormal, -pathRay.direction, ld, albedo, mat.metallic, mat.roughness);
                    float cx = gmax(dot(ld, hit.worldNormal), 0.0f);
                    float cy = gmax(dot(-ld, ln), 1e-12f);
                    float lsa = (1.0f / TriArea(p0, p1, p2)) * (dist * dist) / cy;
                    float pd = sl.pmf * lsa;
                    float pb = BRDFHemispherePDF(hit.worldNormal, -pathRay.direction, ld, albedo, mat.metallic, mat.roughness);
                    vec3 emission = sc.materials[lt.materialIndex].GetEmission();
                    vec3 add = (maxBounces == 1) ? T * brdf * cx * emission / pd : (pd / gmax(pb + pd, 1e-12f)) * T * brdf * cx * emission / pd;
                    dead = add.x == 0.0f && add.y == 0.0f && add.z == 0.0f;
                }
                Payload sh = dead ? Miss() : tracer.TraceTo(shadow, sl.emitterIndex, c);
                if (sh.hitDistance > 0.0f && (uint32_t)sh.objectIndex == sl.emitterIndex) {
                    vec3 ln = TriNormal(n0, n1, n2);
                    vec3 brdf = CalculateBRDF(hit.worldNormal, -pathRay.direction, ld, albedo, mat.metallic, mat.roughness);
                    float cx = gmax(dot(ld, hit.worldNormal), 0.0f);
                    float cy = gmax(dot(-ld, ln), 1e-12f);
                    float triAreaPDF = 1.0f / TriArea(p0, p1, p2);
                    float lightSolidAnglePDF = triAreaPDF * (dist * dist) / cy;
                    pdfDirect = sl.pmf * lightSolidAnglePDF;
                    pdfBRDF = BRDFHemispherePDF(hit.worldNormal, -pathRay.direction, ld, albedo, mat.metallic, mat.roughness);
                    vec3 emission = sc.materials[lt.materialIndex].GetEmission();
                    if (maxBounces == 1) { radiance += T * brdf * cx * emission / pdfDirect; break; }
                    float wD = pdfDirect / gmax(pdfBRDF + pdfDirect, 1e-12f);
                    radiance += wD * T * brdf * cx * emission / pdfDirect;
                }
                if (maxBounces == 1) break;
                vec3 nd = BRDFSampleHemisphere(hit.worldNormal, -pathRay.direction, albedo, mat.metallic, mat.roughness, seed, pdfBRDF);
                pdfBRDF = gmax(pdfBRDF, 1e-12f);
                vec3 brdf = CalculateBRDF(hit.worldNormal, -pathRay.direction, nd, albedo, mat.metallic, mat.roughness);
                float cosT = dot(nd, hit.worldNormal);
                T *= brdf * cosT / pdfBRDF;
                pathRay.origin = hit.worldPosition + hit.worldNormal * 1e-12f; pathRay.direction = nd;
                hit = tracer.Trace(pathRay, c);
                if (hit.hitDistance < 0.0f) { radiance += T * st.skyColor; break; }
                const TriIdx& et = sc.triangles[hit.objectIndex];
                const Material& em = sc.materials[et.materialIndex];
                vec3 emission = em.GetEmission();
                if (em.GetEmissionRadiance() > 0.0f) {
                    sp.normal = hit.worldNormal; sp.position = hit.worldPosition;
                    p0 = sc.worldVertices[et.v0].position; p1 = sc.worldVertices[et.v1].position; p2 = sc.worldVertices[et.v2].position;
                    n0 = sc.worldVertices[et.v0].normal; n1 = sc.worldVertices[et.v1].normal; n2 = sc.worldVertices[et.v2].normal;
                    lp = TriRandomPoint(p0, p1, p2, seed);
                    ld = lp - hit.worldPosition; dist = length(ld); ld /= dist;
                    float cy = gmax(dot(-ld, TriNormal(n0, n1, n2)), 1e-12f);
                    float triAreaPDF = 1.0f / TriArea(p0, p1, p2);
                    float lightSolidAnglePDF = triAreaPDF * (dist * dist) / cy;
                    pdfDirect = ComputeDirectEmitterPMF(sc, sp, (uint32_t)hit.objectIndex) * lightSolidAnglePDF;
                    float wB = pdfBRDF / gmax(pdfBRDF + pdfDirect, 1e-12f);
                    radiance += wB * T * emission;
                    break;
                }
            }
        }
        return v4(radiance / (float)sampleCount, 1.0f);
    }

    // ---- unshadowed target p̂ of emissive-list entry k at the primary hit, evaluated at the
    //      light's centroid (Renderer.cu:1680-1730 and :1799-1849)
    float DI_TargetPdf(uint32_t k, const Payload& pp, vec3 primaryDir, const Material& hm) const {
        uint32_t ti = sc.emissiveTriangles[k];
        const TriIdx& et = sc.triangles[ti];
        const Vertex &a = sc.worldVertices[et.v0], &b = sc.worldVertices[et.v1], &cc = sc.worldVertices[et.v2];
        vec3 ep = TriCentroid(a.position, b.position, cc.position);
        vec3 dir = ep - pp.worldPosition;
        float dist = distance(ep, pp.worldPosition);
        dir = dir / dist;
        vec3 albedo = Albedo(hm, pp.u, pp.v);
        vec3 brdf = CalculateBRDF(pp.worldNormal, -primaryDir, dir, albedo, hm.metallic, hm.roughness);
        const Material& em = sc.materials[et.materialIndex];
        float cx = gmax(dot(dir, pp.worldNormal), 0.0f);
        float cy = gmax(dot(-dir, TriNormal(a.normal, b.normal, cc.normal)), 0.0f);
        float triAreaPDF = 1.0f / TriArea(a.position, b.position, cc.position);
        float solidAnglePDF = triAreaPDF * (dist * dist);
        vec3 Lr = brdf * cx * cy / solidAnglePDF * em.GetEmission();
        return length(Lr);
    }
    // reprojection to the previous frame's pixel (Renderer.cu:1750-1763) + R5
    uint32_t PrevPixel(vec3 worldPos, uint32_t& row) const {
        vec2 uvPrev = GetUVFromNDC(cam.prevProjection, cam.prevView, worldPos);
        vec2 viewport{(float)cam.width, (float)cam.height};
        vec2 sp = uvPrev * viewport;
        auto toInt = [](float f) -> int { if (!(f == f)) return 0; if (f >= 2147483520.0f) return 2147483647; if (f <= -2147483648.0f) return (-2147483647 - 1); return (int)f; };
        int px = iclamp(toInt(floorf(sp.x)), 0, (int)cam.width - 1), py = iclamp(toInt(floorf(sp.y)), 0, (int)cam.height - 1);
        row = (uint32_t)py;
        return (uint32_t)py * cam.width + (uint32_t)px;
    }
    // spatial neighbour pick (Renderer.cu:1915-1922): unsigned wrap of x + int(...) included
    uint32_t NeighborIndex(uint32_t x, uint32_t y, uint8_t radius, uint32_t& seed) const {
        float ox = 2.0f * randomFloat(seed) - 1.0f, oy = 2.0f * randomFloat(seed) - 1.0f;
        ox = (float)(uint32_t)(x + (uint32_t)(int)(ox * (float)radius));
        oy = (float)(uint32_t)(y + (uint32_t)(int)(oy * (float)radius));
        ox = fmaxf(0.0f, fminf((float)cam.width - 1.0f, ox));
        oy = fmaxf(0.0f, fminf((float)cam.height - 1.0f, oy));
        return (uint32_t)ox + (uint32_t)oy * fr.W;
    }

    // ---- ReSTIR DI Part 1 (Renderer.cu:1628-1873)
    vec4 DI_Part1(uint32_t x, uint32_t y, Counters& c) {
        const uint32_t i = x + y * fr.W;
        uint32_t seed = i; seed *= fr.frameIndex + 1 + st.randSeed;
        Ray primary{cam.position, RayDirection(x, y)};
        Payload pp = tracer.Trace(primary, c);
        fr.payload[i] = pp;
        DIReservoir& R = fr.di[i]; DI_Reset(R);
        fr.normalCur[i] = EncodeOctahedral(pp.worldNormal);                      // R2
        if (pp.hitDistance < 0.0f) { fr.depth[i] = pp.hitDistance; return v4(st.skyColor, 1.0f); }
        const Material& hm = MatOf(pp.objectIndex);
        if (length(hm.GetEmission()) > 0.0f) { fr.depth[i] = pp.hitDistance; return v4(hm.GetEmission(), 1.0f); }
        const uint32_t nE = (uint32_t)sc.emissiveTriangles.size();
        const uint32_t candidateCount = (uint32_t)st.lightCandidateCount;
        for (uint32_t k = 0; k < candidateCount; ++k) {
            uint32_t e = (uint32_t)roundf((float)(nE - 1) * randomFloat(seed));
            float pdf = DI_TargetPdf(e, pp, primary.direction, hm);
            float weight = pdf * (float)nE;
            DI_Update(R, e, weight, 1, pdf, seed);
        }
        R.weightEmissive = R.emissivePDF > 0.0f ? (1.0f / R.emissivePDF) * R.weightSum / (float)R.M : 0.0f;
        if (st.useTemporalReuse) {
            uint32_t prow; uint32_t prevIdx = PrevPixel(pp.worldPosition, prow);
            vec3 prevNormal = DecodeOctahedral(fr.normalPrev[prevIdx]);
            DIReservoir prev = fr.diPrev[prevIdx];                               // R3: local copy
            bool validHistory = (double)dot(prevNormal, pp.worldNormal) >= 0.99 && prow >= fr.histDI[0] && prow < fr.histDI[1] && prev.indexEmissive < nE;   // + R7
            DIReservoir T; DI_Reset(T);
            if (validHistory && prev.M > 0) {
                uint8_t historyLimit = (uint8_t)st.temporalHistoryLimit;
                uint32_t lim = (uint32_t)historyLimit * R.M;
                prev.M = (lim < prev.M) ? lim : prev.M;
                uint32_t Z = 0;
                { float pdf = R.emissivePDF; DI_Update(T, R.indexEmissive, pdf * R.weightEmissive * (float)R.M, R.M, pdf, seed); Z += pdf > 0.0f ? R.M : 0; }
                float pdf = DI_TargetPdf(prev.indexEmissive, pp, primary.direction, hm);
                DI_Update(T, prev.indexEmissive, pdf * prev.weightEmissive * (float)prev.M, prev.M, pdf, seed);
                Z += pdf > 0.0f ? prev.M : 0;
                float m = 1.0f / (float)Z;
                T.weightEmissive = T.emissivePDF > 0.0f ? (1.0f / T.emissivePDF) * (m * T.weightSum) : 0.0f;
                R = T;
            }
        }
        return vec4{0, 0, 0, 0};
    }
    // ---- ReSTIR DI Part 2 (Renderer.cu:1875-2041)
    vec4 DI_Part2(uint32_t x, uint32_t y, Counters& c) {
        const uint32_t i = x + y * fr.W;
        uint32_t seed = i; seed *= fr.frameIndex + 1 * 213 + st.randSeed;
        vec3 radiance = v3(0.0f);
        DIReservoir R = fr.di[i];
        Payload pp = fr.payload[i];
        const Material& hm = MatOf(pp.objectIndex);
        vec3 primaryDir = RayDirection(x, y);
        if (st.useSpatialReuse) {
            uint8_t numNeighbors = (uint8_t)st.spatialNeighborNum, radius = (uint8_t)st.spatialNeighborRadius;
            uint32_t Z = 0; DIReservoir S; DI_Reset(S);
            { float pdf = R.emissivePDF; DI_Update(S, R.indexEmissive, pdf * R.weightEmissive * (float)R.M, R.M, pdf, seed); Z += pdf > 0.0f ? R.M : 0; }
            for (uint8_t n = 0; n < numNeighbors; ++n) {
                uint32_t ni = NeighborIndex(x, y, radius, seed);
                float nd = fr.payload[ni].hitDistance, pd = pp.hitDistance;
                if ((nd > 1.1f * pd || nd < 0.9f * pd) || (double)dot(pp.worldNormal, DecodeOctahedral(fr.normalCur[ni])) < 0.906) continue;
                DIReservoir N = fr.di[ni];
                float pdf = N.emissivePDF;
                DI_Update(S, N.indexEmissive, pdf * N.weightEmissive * (float)N.M, N.M, pdf, seed);
                Z += pdf > 0.0f ? N.M : 0;
            }
            float m = 1.0f / (float)Z;
            S.weightEmissive = S.emissivePDF > 0.0f ? (1.0f / S.emissivePDF) * (m * S.weightSum) : 0.0f;
            R = S;
        }
        uint32_t ti = sc.emissiveTriangles[R.indexEmissive];
        const TriIdx& et = sc.triangles[ti];
        const Vertex &a = sc.worldVertices[et.v0], &b = sc.worldVertices[et.v1], &cc = sc.worldVertices[et.v2];
        vec3 ep = TriRandomPoint(a.position, b.position, cc.position, seed);
        vec3 dir = ep - pp.worldPosition;
        float dist = distance(ep, pp.worldPosition);
        dir = dir / dist;
        vec3 albedo = Albedo(hm, pp.u, pp.v);
        vec3 brdf = CalculateBRDF(pp.worldNormal, -primaryDir, dir, albedo, hm.metallic, hm.roughness);
        float cx = gmax(dot(dir, pp.worldNormal), 0.0f);
        float cy = gmax(dot(-dir, TriNormal(a.normal, b.normal, cc.normal)), 0.0f);
        float triAreaPDF = 1.0f / TriArea(a.position, b.position, cc.position);
        float solidAnglePDF = triAreaPDF * (dist * dist);
        vec3 T = brdf * cx * cy / solidAnglePDF;
        Ray ray{pp.worldPosition + pp.worldNormal * 1e-12f, dir};
        // product twin only: the product does not trace a shadow ray whose pixel is black in every outcome (both candidate radiances
        // exactly zero) — the pixel is the reference's, the ray counters are the product's
        bool dead = false;
        if (skipDeadShadowRays) {
            vec3 Lvis = v3(0.0f);
            const Material& m = sc.materials[et.materialIndex];
            if (m.GetEmissionRadiance() > 0.0f) { Lvis = T * m.GetEmission(); Lvis *= R.weightEmissive; }
            const vec3 Lsky = T * st.skyColor;
            auto zero = [](const vec3& v) { return v.x == 0.0f && v.y == 0.0f && v.z == 0.0f; };
            dead = zero(Lvis) && zero(Lsky);
        }
        if (!dead) {
        Payload hit = tracer.TraceTo(ray, ti, c);
        bool vis = (uint32_t)hit.objectIndex == ti;
        if (vis && hit.hitDistance >= 0.0f) {
            const Material& m = sc.materials[et.materialIndex];
            if (m.GetEmissionRadiance() > 0.0f) { radiance = T * m.GetEmission(); radiance *= R.weightEmissive; }
        } else if (hit.hitDistance < 0.0f) radiance = T * st.skyColor;
        }
        fr.depth[i] = pp.hitDistance;
        fr.diPrev[i] = R;
        return v4(radiance, 1.0f);
    }

    // ---- ReSTIR GI Part 1 (Renderer.cu:2043-2293)
    vec4 GI_Part1(uint32_t x, uint32_t y, Counters& c) {
        const uint8_t maxBounces = (uint8_t)st.lightBounces;
        const uint32_t i = x + y * fr.W;
        uint32_t seed = i; seed *= fr.frameIndex + 1 + st.randSeed;
        Ray primary{cam.position, RayDirection(x, y)};
        Payload pp = tracer.Trace(primary, c);
        fr.payload[i] = pp;
        GIReservoir& R = fr.gi[i]; GI_Reset(R);
        fr.normalCur[i] = EncodeOctahedral(pp.worldNormal);                      // R2
        if (pp.hitDistance < 0.0f) { fr.depth[i] = pp.hitDistance; return v4(st.skyColor, 1.0f); }
        const Material& hm = MatOf(pp.objectIndex);
        if (length(hm.GetEmission()) > 0.0f) { fr.depth[i] = pp.hitDistance; return v4(hm.GetEmission(), 1.0f); }
        {
            uint32_t originalSeed = seed;
            vec3 T = v3(1.0f), Lo = v3(0.0f), samplePoint = v3(0.0f), sampleNormal = v3(0.0f);
            Payload hit = pp;
            vec3 albedo = Albedo(hm, hit.u, hit.v);
            float pdf;
            vec3 dir = BRDFSampleHemisphere(pp.worldNormal, -primary.direction, albedo, hm.metallic, hm.roughness, seed, pdf);
            vec3 brdf = CalculateBRDF(pp.worldNormal, -primary.direction, dir, albedo, hm.metallic, hm.roughness);
            float cosT = gmax(dot(dir, pp.worldNormal), 0.0f);
            T *= brdf * cosT / pdf;
            Ray ray{pp.worldPosition + pp.worldNormal * 1e-12f, dir};
            for (int b = 0; b < (int)maxBounces; ++b) {
                seed += (uint32_t)(31 * b);
                hit = tracer.Trace(ray, c);
                if (b == 0) { samplePoint = hit.worldPosition; sampleNormal = hit.worldNormal; }
                if (hit.hitDistance < 0.0f) { Lo += T * st.skyColor; break; }
                const Material& m = MatOf(hit.objectIndex);
                vec3 em = m.GetEmission();
                if (length(em) > 0.0f) { Lo += T * em; break; }
                vec3 alb = Albedo(m, hit.u, hit.v);
                float bpdf;
                vec3 bdir = BRDFSampleHemisphere(hit.worldNormal, -ray.direction, alb, m.metallic, m.roughness, seed, bpdf);
                vec3 bbrdf = CalculateBRDF(hit.worldNormal, -ray.direction, bdir, alb, m.metallic, m.roughness);
                float bcos = gmax(dot(bdir, hit.worldNormal), 0.0f);
                T *= bbrdf * bcos / bpdf;
                ray.origin = hit.worldPosition + hit.worldNormal * 1e-12f; ray.direction = bdir;
            }
            GISample s; s.randSeed = originalSeed; s.visiblePoint = pp.worldPosition; s.visibleNormal = EncodeOctahedral(pp.worldNormal);
            s.samplePoint = samplePoint; s.sampleNormal = EncodeOctahedral(sampleNormal); s.Lo = Lo; s.samplePDF = 0.0f;
            float len = length(s.Lo);
            GI_Update(R, s, len, 1, len, seed);
            R.weightSample = R.sample.samplePDF > 0.0f ? (1.0f / R.sample.samplePDF) * R.weightSum / (float)R.M : 0.0f;
        }
        if (st.useTemporalReuse) {
            uint32_t prow; uint32_t prevIdx = PrevPixel(pp.worldPosition, prow);
            vec3 prevNormal = DecodeOctahedral(fr.normalPrev[prevIdx]);
            GIReservoir prev = fr.giPrev[prevIdx];
            bool validHistory = (double)dot(prevNormal, pp.worldNormal) >= 0.99 && prow >= fr.histGI[0] && prow < fr.histGI[1];   // + R7
            GIReservoir T = R;
            if (validHistory && GI_Valid(prev)) {
                uint8_t historyLimit = (uint8_t)st.temporalHistoryLimit;
                uint32_t lim = (uint32_t)historyLimit * R.M;
                prev.M = (lim < prev.M) ? lim : prev.M;
                float pdf = length(prev.sample.Lo);
                GI_Update(T, prev.sample, pdf * prev.weightSample * (float)prev.M, prev.M, pdf, seed);
                T.weightSample = T.sample.samplePDF > 0.0f ? T.sample.samplePDF / ((float)T.M * T.sample.samplePDF) : 0.0f;
                GI_Reset(R);
                GI_Merge(R, T, T.sample.samplePDF, seed);
            }
        }
        return vec4{0, 0, 0, 0};
    }
    // ---- ReSTIR GI Part 2 (Renderer.cu:2295-2387)
    vec4 GI_Part2(uint32_t x, uint32_t y, Counters& c) {
        const uint32_t i = x + y * fr.W;
        GIReservoir R = fr.gi[i];
        const Payload& pp = fr.payload[i];
        uint32_t seed = i; seed *= fr.frameIndex + 1 * 213 + st.randSeed;
        if (st.useSpatialReuse) {
            uint8_t numNeighbors = (uint8_t)st.spatialNeighborNum, radius = (uint8_t)st.spatialNeighborRadius;
            float plen = length(R.sample.Lo);
            uint32_t Z = plen > 0.0f ? R.M : 0;
            for (uint8_t n = 0; n < numNeighbors; ++n) {
                uint32_t ni = NeighborIndex(x, y, radius, seed);
                float nd = fr.payload[ni].hitDistance, pd = pp.hitDistance;
                GIReservoir N = fr.gi[ni];
                float nlen = length(N.sample.Lo);
                if ((nd > 1.1f * pd || nd < 0.9f * pd) || (double)dot(pp.worldNormal, DecodeOctahedral(fr.normalCur[ni])) < 0.906 || nlen == 0.0f) continue;
                Z += N.M;
                vec3 sn = DecodeOctahedral(N.sample.sampleNormal);
                vec3 dQ = normalize(N.sample.visiblePoint - N.sample.samplePoint);
                float cosQ = dot(sn, dQ);
                vec3 dR = normalize(R.sample.visiblePoint - N.sample.samplePoint);
                float cosR = dot(sn, dR);
                float jl = cosQ > 0.0f ? cosR / cosQ : 0.0f;
                float distQ = length(N.sample.visiblePoint - N.sample.samplePoint), distR = length(R.sample.visiblePoint - N.sample.samplePoint);
                float jr = distR > 0.0f ? (distQ * distQ) / (distR * distR) : 0.0f;
                float jac = jl * jr;
                float pdf = jac > 0.0f ? nlen / jac : 0.0f;
                Ray ray{N.sample.samplePoint, normalize(R.sample.visiblePoint - N.sample.samplePoint)};
                float dist = length(R.sample.visiblePoint - N.sample.samplePoint);
                float tol = gmax(1e-4f, dist * 1e-3f);
                // (product twin: a visibility ray whose merge weight is zero already is not traced — same reservoir, the product's ray count)
                bool visible = (skipDeadShadowRays && pdf == 0.0f) ? true : tracer.TraceVisible(ray, dist, tol, c);
                if (!visible) pdf = 0.0f;
                GI_Merge(R, N, pdf, seed);
            }
            R.weightSample = R.sample.samplePDF > 0.0f ? R.sample.samplePDF / ((float)Z * R.sample.samplePDF) : 0.0f;
        }
        vec3 radiance = R.sample.Lo * R.weightSample;
        fr.depth[i] = pp.hitDistance;
        fr.giPrev[i] = R;
        return v4(radiance, 1.0f);
    }

    // ---- common epilogue "E" (e.g. Renderer.cu:2448-2465)
    void Epilogue(uint32_t i, vec4 c) {
        if (!finite4(c)) c = vec4{0, 0, 0, 0};
        fr.accum[i] = fr.accum[i] + c;
        vec4 a = fr.accum[i] / (float)fr.frameIndex;
        a = a / (a + vec4{1, 1, 1, 0});
        a = vec4{gclamp(a.x, 0.0f, 1.0f), gclamp(a.y, 0.0f, 1.0f), gclamp(a.z, 0.0f, 1.0f), gclamp(a.w, 0.0f, 1.0f)};
        fr.image[i] = ConvertToRGBA(a);
    }

    // One Renderer::Render call (Renderer.cu:13-284) restricted to rows [y0, y1); ReSTIR Part 1
    // additionally covers `halo` rows either side (tile-split halo recompute).  Returns counters.
    Counters RenderFrame(uint32_t y0, uint32_t y1, uint32_t halo = 0) {
        const uint32_t W = fr.W;
        if (fr.frameIndex == 1) for (auto& a : fr.accum) a = vec4{0, 0, 0, 0};   // :50-51
        Counters total;
        const int tech = st.technique;
        const uint32_t h0 = (y0 > halo) ? y0 - halo : 0, h1 = (y1 + halo < fr.H) ? y1 + halo : fr.H;
        auto rows = [&](uint32_t ra, uint32_t rb, auto&& fn) {
            #pragma omp parallel
            {
                Counters local;
                // work items = 64-pixel row segments, dealt one at a time: 32 400 items at 1080p keep a few hundred host threads
                // evenly busy (whole rows in chunks of 4 left half of a 256-thread host idle at the end of every pass)
                const int segs = (int)((W + 63u) / 64u), items = (int)(rb - ra) * segs;
                #pragma omp for schedule(dynamic, 1) nowait
                for (int it = 0; it < items; ++it) {
                    const uint32_t y = ra + (uint32_t)(it / segs), x0 = (uint32_t)(it % segs) * 64u, x1 = (x0 + 64u < W) ? x0 + 64u : W;
                    for (uint32_t x = x0; x < x1; ++x) fn(x, y, local);
                }
                #pragma omp critical
                { total.rays += local.rays; total.boxTests += local.boxTests; total.triTests += local.triTests; total.hits += local.hits; total.nodeVisits += local.nodeVisits; }
            }
        };
        if (tech == RESTIR_DI || tech == RESTIR_GI) {
            auto part1 = [&](uint32_t x, uint32_t y, Counters& c) {
                uint32_t i = x + y * W;
                vec4 col = (tech == RESTIR_DI) ? DI_Part1(x, y, c) : GI_Part1(x, y, c);
                if (y < y0 || y >= y1) return;                                                                           // halo rows: reservoirs only
                if (col.x == 0.0f && col.y == 0.0f && col.z == 0.0f && col.w == 0.0f) fr.image[i] = ConvertToRGBA(col);   // sentinel :2746-2750
                else Epilogue(i, col);
            };
            rows(h0, h1, part1);
            // tile split only: a neighbour offset above row 0 wraps (unsigned, R.cu:1916-1917) and clamps to the LAST row,
            // so a band owning rows < radius also needs Part 1 of row H-1
            if (halo > 0 && y0 < halo && h1 < fr.H) rows(fr.H - 1, fr.H, part1);
            rows(y0, y1, [&](uint32_t x, uint32_t y, Counters& c) {
                uint32_t i = x + y * W;
                vec4 u = UnpackABGR(fr.image[i]);
                if (u.x == 0.0f && u.y == 0.0f && u.z == 0.0f && u.w == 0.0f)      // :2787
                    Epilogue(i, (tech == RESTIR_DI) ? DI_Part2(x, y, c) : GI_Part2(x, y, c));
            });
            fr.normalPrev.swap(fr.normalCur);                                     // R2
            if (tech == RESTIR_DI) { fr.histDI[0] = y0; fr.histDI[1] = y1; } else { fr.histGI[0] = y0; fr.histGI[1] = y1; }   // R7
        } else {
            rows(y0, y1, [&](uint32_t x, uint32_t y, Counters& c) {
                vec4 col;
                if (tech == LIGHT_SOURCE_SAMPLING) col = PerPixel_LightSource(x, y, c);
                else if (tech == NEE) col = PerPixel_NEE(x, y, c);
                else col = PerPixel_Path(x, y, (tech >= BRUTE_FORCE && tech <= BRDF_SAMPLING) ? tech : BRUTE_FORCE, c);
                Epilogue(x + y * W, col);
            });
        }
        if (st.toAccumulate) fr.frameIndex++; else fr.frameIndex = 1;             // :258-261
        return total;
    }
};

}  // namespace orc
