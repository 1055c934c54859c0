// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_vec.h header).
//
// Instrumented CPU restatement of the *product's own* traversal order over the product's
// own acceleration structure (exported as plain arrays through the C ABI,
// include/fyprt.h: fyprt_export_bvh).  SURVEY.md §8(d) defines the algorithmic bytes per
// ray as 32*n_box + 36*n_tri + 40*[hit] with n_box / n_tri "counted by an instrumented CPU
// restatement of that traversal on the identical scene/camera/seed" — this is that
// restatement.  It consumes DATA produced by the product (node / triangle arrays); it
// shares no code with it.  Layout contract:
//   node  (64 B): float origin[3]; uint8 ex[3], count; int32 child[4]; uint8 qlo[3][4], qhi[3][4]; uint32 pad[2]
//                 child plane on axis a = origin[a] + q * 2^(ex[a] - 127); children 0..count-1 valid
//   child >= 0 : inner node index;  child < 0 : leaf, ~child = (firstTri << 2) | (count-1)
//   tri   (48 B): float v0[3], e1[3], e2[3]; uint32 triangleIndex; uint32 pad[2]
// Traversal: per visit A_a = 2^(ex_a-127) * (1/d_a), B_a = (origin_a - o_a) * (1/d_a), slab parameter of a plane
// t = fma(q, A_a, B_a), entry plane = lo where 1/d_a >= 0 else hi (unused slots hold lo 255 / hi 0 and never pass);
// children culled against cut (= closest*1.000001f, or the light / visibility distance), the entry distance clamped to >= 1e-30; hit
// children ordered by the 64-bit key (entry distance bits << 32 | child reference) with the 5-comparator network
// (0,1)(2,3)(0,2)(1,3)(1,2) — equal distances go by reference; nearest visited next, the others pushed far-to-near; |d| < 1e-30 replaced by copysign(1e-30, d);
// Möller–Trumbore identical to Renderer.cu:513-537 on (v0, e1, e2).
#pragma once
#include "oracle_render.h"

namespace orc {

struct PNode { float origin[3]; uint8_t ex[3], meta; int32_t child[4]; uint8_t qlo[3][4], qhi[3][4]; uint32_t pad[2]; };
struct PTri { float v0[3], e1[3], e2[3]; uint32_t tri, pad[2]; };
static_assert(sizeof(PNode) == 64 && sizeof(PTri) == 48, "layout");

struct ProductTracer : Tracer {
    const Scene& sc; std::vector<PNode> nodes; std::vector<PTri> tris; int32_t rootRef = 0;
    explicit ProductTracer(const Scene& s) : sc(s) {}
    static inline float safeInv(float d) { return 1.0f / ((fabsf(d) < 1e-30f) ? copysignf(1e-30f, d) : d); }
    // non-finite rays are answered as misses up front (the product's ray_not_finite; no triangle test can pass for them)
    static inline bool notFinite(const Ray& r) {
        const vec3 &o = r.origin, &d = r.direction;
        const float z = ((o.x - o.x) + (o.y - o.y)) + ((o.z - o.z) + (d.x - d.x)) + ((d.y - d.y) + (d.z - d.z));
        return !(z == 0.0f);
    }
    static inline float pow2e(uint8_t e) { uint32_t b = (uint32_t)e << 23; float f; memcpy(&f, &b, 4); return f; }
    // one node visit (`cur` is a node index or a resume entry): pushes the hit children that are not visited next — one by
    // one, far-to-near, while pending + 2 + levels(node) <= 31, else as ONE resume entry (node << 4 | remaining hit slots) —
    // and returns true and the nearest hit child in `next` if any child is hit.
    static constexpr int32_t kResumeBase = 0x40000000;
    bool Visit(int32_t cur, float ox, float oy, float oz, float ix, float iy, float iz, float cut, int32_t* stack, int& top, int32_t& next, Counters& c) const {
        const bool resumed = cur >= kResumeBase;
        const int32_t node = resumed ? ((cur - kResumeBase) >> 4) : cur;
        const PNode& n = nodes[node];
        const uint32_t cnt = n.meta & 7u, levels = n.meta >> 3;
        const uint32_t allow = resumed ? ((uint32_t)cur & 0xFu) : 0xFu;
        c.boxTests += (uint64_t)__builtin_popcount(allow & ((1u << cnt) - 1u)); c.nodeVisits++;
        const float o[3] = {ox, oy, oz}, inv[3] = {ix, iy, iz};
        float A[3], B[3];
        for (int a = 0; a < 3; ++a) { A[a] = pow2e(n.ex[a]) * inv[a]; B[a] = (n.origin[a] - o[a]) * inv[a]; }
        // sort keys: (bit pattern of the entry distance, child reference) as one unsigned 64-bit number — what the product's
        // v_min_f64 / v_max_f64 network orders (numeric order of positive normal doubles = order of their bit patterns)
        constexpr uint32_t kMissHi = 0x7F900000u;
        uint64_t key[4]; uint32_t slotHi[4];
        for (int i = 0; i < 4; ++i) {
            float tn[3], tf[3];       // entry through the lo plane where the ray travels in +axis direction, else through the hi plane
            for (int a = 0; a < 3; ++a) {
                const bool neg = inv[a] < 0.0f;
                tn[a] = fmaf((float)(neg ? n.qhi[a][i] : n.qlo[a][i]), A[a], B[a]);
                tf[a] = fmaf((float)(neg ? n.qlo[a][i] : n.qhi[a][i]), A[a], B[a]);
            }
            float tnear = fmaxf(fmaxf(tn[0], tn[1]), fmaxf(tn[2], 1e-30f));      // the product's kNearClamp
            float tfar = fminf(fminf(tf[0], tf[1]), fminf(tf[2], cut));
            uint32_t hi; memcpy(&hi, &tnear, 4);
            slotHi[i] = (tnear <= tfar && ((allow >> i) & 1u)) ? hi : kMissHi;   // unused slots (lo 255, hi 0) never pass
            key[i] = ((uint64_t)slotHi[i] << 32) | (uint32_t)n.child[i];
        }
        auto order = [&](int a, int b) { if (key[b] < key[a]) std::swap(key[a], key[b]); };
        order(0, 1); order(2, 3); order(0, 2); order(1, 3); order(1, 2);
        auto hiOf = [&](int i) { return (uint32_t)(key[i] >> 32); };
        auto refOf = [&](int i) { return (int32_t)(uint32_t)key[i]; };
        if (top + 2 + (int)levels <= stackBudget) {
            if (hiOf(3) < kMissHi) stack[top++] = refOf(3);
            if (hiOf(2) < kMissHi) stack[top++] = refOf(2);
            if (hiOf(1) < kMissHi) stack[top++] = refOf(1);
        } else if (hiOf(1) < kMissHi) {
            uint32_t hit = 0, nearest = 8;
            for (int i = 0; i < 4; ++i) if (slotHi[i] < kMissHi) hit |= 1u << i;
            for (int i = 0; i < 4; ++i) if (n.child[i] == refOf(0)) { nearest = 1u << i; break; }
            stack[top++] = kResumeBase + (int32_t)(((uint32_t)node << 4) | (hit & ~nearest));
        }
        maxTop = top > maxTop ? top : maxTop;
        if (hiOf(0) < kMissHi) { next = refOf(0); return true; }
        return false;
    }
    int stackBudget = 31;       // the product's tuning key 8
    mutable int maxTop = 0;     // deepest stack seen (diagnostic; racy across OpenMP threads, only ever compared with 31)
    // Event log for tools/wave_sim.py (scheduling studies of the persistent trace kernels; single-threaded renders only): per ray
    // 0xF0 | kind (0 closest, 1 shadow, 2 visibility), then 0x01 per node visit and 0x10 | n per leaf with n triangle tests executed, 0xFF at its end.
    mutable std::vector<uint8_t>* events = nullptr;
    inline void ev(uint8_t b) const { if (events) events->push_back(b); }
    Payload Trace(const Ray& ray, Counters& c) const override {
        c.rays++;
        if (tris.empty() || notFinite(ray)) return Miss();
        const float ox = ray.origin.x, oy = ray.origin.y, oz = ray.origin.z;
        const float ix = safeInv(ray.direction.x), iy = safeInv(ray.direction.y), iz = safeInv(ray.direction.z);
        float closest = FLT_MAX, closestInfl = closest * 1.000001f; int closestTri = -1; float cu = 0.0f, cv = 0.0f;
        int32_t stack[128]; int top = 0; int32_t cur = rootRef;
        ev(0xF0);
        while (true) {
            if (cur >= 0) {
                ev(0x01);
                if (Visit(cur, ox, oy, oz, ix, iy, iz, closestInfl, stack, top, cur, c)) continue;
            } else {
                if (cur == INT32_MIN) break;                       // the product's exit sentinel (an unused child slot holds it)
                uint32_t code = (uint32_t)~cur; uint32_t first = code >> 2, cnt = (code & 3u) + 1u;
                ev((uint8_t)(0x10 | cnt));
                for (uint32_t k = 0; k < cnt; ++k) {
                    const PTri& T = tris[first + k];
                    c.triTests++;
                    vec3 v0 = v3(T.v0[0], T.v0[1], T.v0[2]), e1 = v3(T.e1[0], T.e1[1], T.e1[2]), e2 = v3(T.e2[0], T.e2[1], T.e2[2]);
                    vec3 h = cross(ray.direction, e2);
                    float a = dot(e1, h), f = 1.0f / a;
                    vec3 s = ray.origin - v0;
                    float u = f * dot(s, h);
                    if (u < 0.0f || u > 1.0f) continue;
                    vec3 q = cross(s, e1);
                    float v = f * dot(ray.direction, q);
                    if (v < 0.0f || (u + v) > 1.0f) continue;
                    float t = f * dot(e2, q);
                    if (t > 0.0001f && t < closest) { closest = t; closestInfl = closest * 1.000001f; closestTri = (int)T.tri; cu = u; cv = v; }
                }
            }
            if (top == 0) break;
            cur = stack[--top];
        }
        ev(0xFF);
        if (closestTri < 0) return Miss();
        c.hits++;
        return ClosestHit(sc, ray, closest, closestTri, cu, cv);
    }
    // Restatement of the product's trace_shadow (rt_device.h): light triangle first, interval cut at t_light,
    // stop at the first closer triangle; full closest-hit fallback when the ray misses the light triangle.
    // Only hitDistance / objectIndex of the result are meaningful.
    Payload TraceTo(const Ray& ray, uint32_t lightTri, Counters& c) const override {
        float tL = -1.0f;
        {
            const TriIdx& T = sc.triangles[lightTri];
            vec3 v0 = sc.worldVertices[T.v0].position, e1 = sc.worldVertices[T.v1].position - v0, e2 = sc.worldVertices[T.v2].position - v0;
            vec3 h = cross(ray.direction, e2);
            float a = dot(e1, h), f = 1.0f / a;
            vec3 s = ray.origin - v0;
            float u = f * dot(s, h);
            if (!(u < 0.0f || u > 1.0f)) {
                vec3 q = cross(s, e1);
                float v = f * dot(ray.direction, q);
                if (!(v < 0.0f || (u + v) > 1.0f)) { float t = f * dot(e2, q); if (t > 0.0001f) tL = t; }
            }
        }
        if (!(tL > 0.0f)) { c.triTests++; return Trace(ray, c); }
        c.rays++; c.triTests++; c.hits++;
        const float ox = ray.origin.x, oy = ray.origin.y, oz = ray.origin.z;
        const float ix = safeInv(ray.direction.x), iy = safeInv(ray.direction.y), iz = safeInv(ray.direction.z);
        const float cut = tL * 1.000001f;
        int32_t stack[128]; int top = 0; int32_t cur = rootRef;
        Payload r = Miss(); r.hitDistance = tL; r.objectIndex = (int32_t)lightTri;
        ev(0xF1); ev((uint8_t)(lightTri & 0xFFu)); ev((uint8_t)((lightTri >> 8) & 0xFFu)); ev((uint8_t)((lightTri >> 16) & 0xFFu));   // + the light triangle (3 bytes, < 0xF0 each is not required: the reader skips them by position)
        while (true) {
            if (cur >= 0) {
                ev(0x01);
                if (Visit(cur, ox, oy, oz, ix, iy, iz, cut, stack, top, cur, c)) continue;
            } else {
                if (cur == INT32_MIN) break;                       // the product's exit sentinel (an unused child slot holds it)
                uint32_t code = (uint32_t)~cur; uint32_t first = code >> 2, cnt = (code & 3u) + 1u;
                const size_t evAt = events ? events->size() : 0; ev(0x10);
                for (uint32_t k = 0; k < cnt; ++k) {
                    const PTri& T = tris[first + k];
                    if (T.tri == lightTri) continue;
                    c.triTests++; if (events) (*events)[evAt]++;
                    vec3 v0 = v3(T.v0[0], T.v0[1], T.v0[2]), e1 = v3(T.e1[0], T.e1[1], T.e1[2]), e2 = v3(T.e2[0], T.e2[1], T.e2[2]);
                    vec3 h = cross(ray.direction, e2);
                    float a = dot(e1, h), f = 1.0f / a;
                    vec3 s = ray.origin - v0;
                    float u = f * dot(s, h);
                    if (u < 0.0f || u > 1.0f) continue;
                    vec3 q = cross(s, e1);
                    float v = f * dot(ray.direction, q);
                    if (v < 0.0f || (u + v) > 1.0f) continue;
                    float t = f * dot(e2, q);
                    if (t > 0.0001f && t < tL) { r.hitDistance = t; r.objectIndex = (int32_t)T.tri; ev(0xFF); return r; }
                }
            }
            if (top == 0) break;
            cur = stack[--top];
        }
        ev(0xFF);
        return r;
    }
    // Restatement of the product's trace_visible (rt_device.h): interval cut at dist + tol, early "not visible" on any hit
    // closer than dist - tol, visible iff a hit lies inside [dist - tol, dist + tol].
    bool TraceVisible(const Ray& ray, float dist, float tol, Counters& c) const override {
        c.rays++;
        bool found = false;
        if (!tris.empty() && !notFinite(ray)) {
            const float ox = ray.origin.x, oy = ray.origin.y, oz = ray.origin.z;
            const float ix = safeInv(ray.direction.x), iy = safeInv(ray.direction.y), iz = safeInv(ray.direction.z);
            const float tLo = dist - tol, tHi = dist + tol, cut = tHi * 1.000001f;
            int32_t stack[128]; int top = 0; int32_t cur = rootRef;
            while (true) {
                if (cur >= 0) {
                    if (Visit(cur, ox, oy, oz, ix, iy, iz, cut, stack, top, cur, c)) continue;
                } else {
                    if (cur == INT32_MIN) break;
                    uint32_t code = (uint32_t)~cur; uint32_t first = code >> 2, cnt = (code & 3u) + 1u;
                    for (uint32_t k = 0; k < cnt; ++k) {
                        const PTri& T = tris[first + k];
                        c.triTests++;
                        vec3 v0 = v3(T.v0[0], T.v0[1], T.v0[2]), e1 = v3(T.e1[0], T.e1[1], T.e1[2]), e2 = v3(T.e2[0], T.e2[1], T.e2[2]);
                        vec3 h = cross(ray.direction, e2);
                        float a = dot(e1, h), f = 1.0f / a;
                        vec3 s = ray.origin - v0;
                        float u = f * dot(s, h);
                        if (u < 0.0f || u > 1.0f) continue;
                        vec3 q = cross(s, e1);
                        float v = f * dot(ray.direction, q);
                        if (v < 0.0f || (u + v) > 1.0f) continue;
                        float t = f * dot(e2, q);
                        if (t > 0.0001f) {
                            if (t < tLo) return false;
                            if (t <= tHi) found = true;
                        }
                    }
                }
                if (top == 0) break;
                cur = stack[--top];
            }
        }
        if (found) c.hits++;
        return found;
    }
};
}  // namespace orc
