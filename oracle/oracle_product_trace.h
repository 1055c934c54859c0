// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_vec.h header).
//
// Instrumented CPU restatement of the *product's own* traversal order over the product's
// own acceleration structure (exported as plain arrays through the C ABI,
// include/fyprt.h: fyprt_export_bvh).  SURVEY.md §8(d) defines the algorithmic bytes per
// ray as 32*n_box + 36*n_tri + 40*[hit] with n_box / n_tri "counted by an instrumented CPU
// restatement of that traversal on the identical scene/camera/seed" — this is that
// restatement.  It consumes DATA produced by the product (node / triangle arrays); it
// shares no code with it.  Layout contract:
//   node  (64 B): float lo0[3], hi0[3], lo1[3], hi1[3]; int32 child0, child1; int32 pad[2]
//   child >= 0 : inner node index;  child < 0 : leaf, ~child = (firstTri << 2) | (count-1)
//   tri   (48 B): float v0[3], e1[3], e2[3]; uint32 triangleIndex; uint32 pad[2]
// Traversal: ordered (near child first, ties -> child0), far child pushed, boxes culled
// against closest*1.000001f, slab test (plane - o) * (1/d) with |d| < 1e-30 replaced by
// copysign(1e-30, d); Möller–Trumbore identical to Renderer.cu:513-537 on (v0, e1, e2).
#pragma once
#include "oracle_render.h"

namespace orc {

struct PNode { float lo0[3], hi0[3], lo1[3], hi1[3]; int32_t child0, child1, pad[2]; };
struct PTri { float v0[3], e1[3], e2[3]; uint32_t tri, pad[2]; };
static_assert(sizeof(PNode) == 64 && sizeof(PTri) == 48, "layout");

struct ProductTracer : Tracer {
    const Scene& sc; std::vector<PNode> nodes; std::vector<PTri> tris; int32_t rootRef = 0;
    explicit ProductTracer(const Scene& s) : sc(s) {}
    static inline float safeInv(float d) { return 1.0f / ((fabsf(d) < 1e-30f) ? copysignf(1e-30f, d) : d); }
    Payload Trace(const Ray& ray, Counters& c) const override {
        c.rays++;
        if (tris.empty()) return Miss();
        const float ox = ray.origin.x, oy = ray.origin.y, oz = ray.origin.z;
        const float ix = safeInv(ray.direction.x), iy = safeInv(ray.direction.y), iz = safeInv(ray.direction.z);
        float closest = FLT_MAX, closestInfl = closest * 1.000001f; int closestTri = -1; float cu = 0.0f, cv = 0.0f;
        int32_t stack[128]; int top = 0; int32_t cur = rootRef;
        auto slab = [&](const float* lo, const float* hi, float& tn) -> bool {
            float ax = (lo[0] - ox) * ix, bx = (hi[0] - ox) * ix;
            float ay = (lo[1] - oy) * iy, by = (hi[1] - oy) * iy;
            float az = (lo[2] - oz) * iz, bz = (hi[2] - oz) * iz;
            float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
            float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), closestInfl));
            tn = tnear; return tnear <= tfar;
        };
        while (true) {
            if (cur >= 0) {
                const PNode& n = nodes[cur];
                c.boxTests += 2;
                float t0, t1; bool h0 = slab(n.lo0, n.hi0, t0), h1 = slab(n.lo1, n.hi1, t1);
                if (h0 && h1) {
                    if (t1 < t0) { stack[top++] = n.child0; cur = n.child1; } else { stack[top++] = n.child1; cur = n.child0; }
                    continue;
                } else if (h0) { cur = n.child0; continue; }
                else if (h1) { cur = n.child1; continue; }
            } else {
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
                    if (t > 0.0001f && t < closest) { closest = t; closestInfl = closest * 1.000001f; closestTri = (int)T.tri; cu = u; cv = v; }
                }
            }
            if (top == 0) break;
            cur = stack[--top];
        }
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
        auto slab = [&](const float* lo, const float* hi, float& tn) -> bool {
            float ax = (lo[0] - ox) * ix, bx = (hi[0] - ox) * ix;
            float ay = (lo[1] - oy) * iy, by = (hi[1] - oy) * iy;
            float az = (lo[2] - oz) * iz, bz = (hi[2] - oz) * iz;
            float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
            float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), cut));
            tn = tnear; return tnear <= tfar;
        };
        Payload r = Miss(); r.hitDistance = tL; r.objectIndex = (int32_t)lightTri;
        while (true) {
            if (cur >= 0) {
                const PNode& n = nodes[cur];
                c.boxTests += 2;
                float t0, t1; bool h0 = slab(n.lo0, n.hi0, t0), h1 = slab(n.lo1, n.hi1, t1);
                if (h0 && h1) { if (t1 < t0) { stack[top++] = n.child0; cur = n.child1; } else { stack[top++] = n.child1; cur = n.child0; } continue; }
                else if (h0) { cur = n.child0; continue; }
                else if (h1) { cur = n.child1; continue; }
            } else {
                uint32_t code = (uint32_t)~cur; uint32_t first = code >> 2, cnt = (code & 3u) + 1u;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const PTri& T = tris[first + k];
                    if (T.tri == lightTri) continue;
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
                    if (t > 0.0001f && t < tL) { r.hitDistance = t; r.objectIndex = (int32_t)T.tri; return r; }
                }
            }
            if (top == 0) break;
            cur = stack[--top];
        }
        return r;
    }
    // Restatement of the product's trace_visible (rt_device.h): interval cut at dist + tol, early "not visible" on any hit
    // closer than dist - tol, visible iff a hit lies inside [dist - tol, dist + tol].
    bool TraceVisible(const Ray& ray, float dist, float tol, Counters& c) const override {
        c.rays++;
        bool found = false;
        if (!tris.empty()) {
            const float ox = ray.origin.x, oy = ray.origin.y, oz = ray.origin.z;
            const float ix = safeInv(ray.direction.x), iy = safeInv(ray.direction.y), iz = safeInv(ray.direction.z);
            const float tLo = dist - tol, tHi = dist + tol, cut = tHi * 1.000001f;
            int32_t stack[128]; int top = 0; int32_t cur = rootRef;
            auto slab = [&](const float* lo, const float* hi, float& tn) -> bool {
                float ax = (lo[0] - ox) * ix, bx = (hi[0] - ox) * ix;
                float ay = (lo[1] - oy) * iy, by = (hi[1] - oy) * iy;
                float az = (lo[2] - oz) * iz, bz = (hi[2] - oz) * iz;
                float tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
                float tfar = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), cut));
                tn = tnear; return tnear <= tfar;
            };
            while (true) {
                if (cur >= 0) {
                    const PNode& n = nodes[cur];
                    c.boxTests += 2;
                    float t0, t1; bool h0 = slab(n.lo0, n.hi0, t0), h1 = slab(n.lo1, n.hi1, t1);
                    if (h0 && h1) { if (t1 < t0) { stack[top++] = n.child0; cur = n.child1; } else { stack[top++] = n.child1; cur = n.child0; } continue; }
                    else if (h0) { cur = n.child0; continue; }
                    else if (h1) { cur = n.child1; continue; }
                } else {
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
