// Wavefront path engine: techniques 0-6 (brute force, uniform, cosine, GGX, BRDF, light-source sampling, NEE + MIS) and
// ReSTIR GI Part 1 / Part 2 as STAGES instead of one-thread-per-pixel megakernels:
//
//   k_primary / k_gi_primary   one thread per pixel, wave = 8x8 tile: primary ray (coherent, traversal stack in LDS), payload;
//                              pixels that saw the sky or an emitter are finished on the spot, the others are appended to a list
//   k_shade<TECH>, step k      one thread per LIVE path, no traversal state at all: consumes the results of the rays the path
//                              emitted in step k-1 (rt_wavefront.h ray records), runs the technique's shading / sampling code up
//                              to the next point where it needs a ray, emits that ray (or two: NEE's shadow + bounce ray) into
//                              the next list — compacted with wave ballot + prefix popcount, an LDS prefix over the 4 waves and
//                              ONE atomic per workgroup — or finishes the pixel (fused epilogue)
//   k_trace_rays               persistent waves over the emitted rays (closest hit / shadow / visibility), rt_wavefront.h
//
// so that no kernel holds traversal state and shading state at once (the megakernels spilled 24-132 VGPRs each) and SIMD lanes
// are re-packed after every bounce.  The reference's loops (R.cu:565-1284 sample x bounce, :1287-1408 light-source samples,
// :1411-1626 NEE sample x bounce, :2043-2293 GI bounce loop, :2295-2387 GI neighbour loop) become state machines: the per-pixel
// state between two steps lives in a small record (`PathIO::state`), everything that is a pure function of the primary hit is
// recomputed.  The arithmetic — every expression, its operation order, the order of the random numbers drawn from the per-pixel
// seed, the order in which contributions are added to the radiance — is the megakernels' (and the reference's), so every output
// bit is unchanged; tests/test_gpu_parity.py compares all of it with the oracle.
#pragma once
#include "rt_wavefront.h"

namespace rt {

enum { T_GI1 = 9, T_GI2 = 10 };          // stage ids beyond the SamplingTechniqueEnum values (Tech)

struct PathIO {
    const uint32_t* pixelList;           // step 0: the owners come from this list (written by the primary kernel / by GI Part 1)
    const float4* raysIn; const float4* hitsIn; const uint32_t* countIn;     // rays emitted by the previous step, their results, number of ENTRIES
    float4* raysOut; uint32_t* countOut;                                      // rays of the next step
    float4* state; uint32_t stateStride;                                      // per-pixel path state, float4 units
    uint32_t iteration, raysPer;                                              // rays per entry (2 for NEE with more than one bounce)
    uint32_t* part2List; uint32_t* part2Count;                                // ReSTIR GI: pixels that continue into Part 2; NEE: the PICK list (counter = countOut)
    uint32_t* misList; uint32_t* misCount;                                    // NEE: paths whose BRDF ray hit an emitter
    const uint32_t* ownersIn; uint32_t* ownersOut;                            // ReSTIR GI Part 2: the entries' pixels as a list of their own (its steps do not read the ray records)
    uint32_t fusedOwner;                                                      // k_path_fused: the thread's own pixel (step 0 has no list to read it from); kNotFused otherwise
    float4* local;                                                            // k_path_fused (LOCAL instantiations of the step functions): the thread's own state (2 quads), ray (3) and hit (1) — registers, not memory
};
constexpr uint32_t kNotFused = 0xFFFFFFFFu;

struct RayRec { float4 q0, q1, q2; };
RT_DEV RayRec ray_closest(f3 o, f3 d, uint32_t pixel) {
    RayRec r; r.q0 = make_float4(o.x, o.y, o.z, __int_as_float((int)pixel)); r.q1 = make_float4(d.x, d.y, d.z, __int_as_float((int)kRayClosest));
    r.q2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); return r;
}
RT_DEV RayRec ray_shadow(const DevScene& sc, f3 o, f3 d, uint32_t pixel, uint32_t lightTri) {
    RayRec r; r.q0 = make_float4(o.x, o.y, o.z, __int_as_float((int)pixel)); r.q1 = make_float4(d.x, d.y, d.z, __int_as_float((int)lightTri));
    r.q2 = make_float4(light_tri_distance(sc, lightTri, o, d), 0.0f, 0.0f, 0.0f); return r;
}
RT_DEV RayRec ray_visible(f3 o, f3 d, uint32_t pixel, float dist, float tol) {
    RayRec r; r.q0 = make_float4(o.x, o.y, o.z, __int_as_float((int)pixel)); r.q1 = make_float4(d.x, d.y, d.z, __int_as_float((int)kRayVisible));
    r.q2 = make_float4(dist, tol, 0.0f, 0.0f); return r;
}
RT_DEV void store_ray(float4* rays, uint32_t index, const RayRec& r) { float4* p = rays + (size_t)index * 3; p[0] = r.q0; p[1] = r.q1; p[2] = r.q2; }
RT_DEV Hit load_hit(const float4* hits, uint32_t index) { const float4 q = hits[index]; Hit h; h.t = q.x; h.u = q.y; h.v = q.z; h.tri = __float_as_int(q.w); return h; }
RT_DEV f3 xyz(float4 q) { return mk3(q.x, q.y, q.z); }
RT_DEV float4 f3w(f3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __int_as_float((int)w)); }
RT_DEV float4 f3f(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

// Slot of this thread's new list entry: ballot + prefix popcount inside a wave, a 4-entry LDS prefix across the waves of the
// workgroup, ONE atomic on the list counter per workgroup.  Every thread of the workgroup must call it.
RT_DEV uint32_t block_append(bool live, uint32_t* counter) {
    __shared__ uint32_t s_cnt[kBlock / 64];
    __shared__ uint32_t s_base;
    const unsigned long long mask = __ballot(live);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0u) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    const uint32_t n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    for (uint32_t k = 0; k < wave; ++k) rank += s_cnt[k];
    if (threadIdx.x == 0u) s_base = n ? atomicAdd(counter, n) : 0u;
    __syncthreads();
    const uint32_t slot = s_base + rank;
    __syncthreads();                                   // s_cnt / s_base are written again by the next call
    return slot;
}

// ============================================================ primary rays of techniques 0-6
// First lines of every PerPixel_* function (e.g. R.cu:565-600): primary ray, sky / directly visible emitter -> finished.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_primary(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st, uint32_t* pixelList, uint32_t* listCount) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch
    uint32_t x, y;
    bool inside = pixel_of_thread(fr, fr.rowBegin, fr.rowEnd, x, y);
    if (fr.stripeRows != 0u) {                                   // [rowBegin, rowEnd) counts this context's rows, stripe after stripe: local row -> image row
        const uint32_t k = y / fr.stripeRows;
        y = (k * fr.stripeParts + fr.stripePart) * fr.stripeRows + (y - k * fr.stripeRows);
        inside = inside && y < fr.H;
    }
    const uint32_t i = x + y * fr.W;
    bool live = false;
    if (inside) {
        const f3 pd = ray_direction(cam, x, y);
        const Payload pp = trace_ray<COUNT>(sc, cam.position, pd, s_stack + threadIdx.x);
        fr.payload[i] = pp;
        if (pp.hitDistance < 0.0f) epilogue(fr, i, rgb1(st.sky));
        else {
            const Mat hm = load_mat(sc, tri_material(sc, pp.objectIndex));
            if (length(emission(hm)) > 0.0f) epilogue(fr, i, rgb1(emission(hm))); else live = true;
        }
    }
    const uint32_t slot = block_append(live, listCount);
    if (live) pixelList[slot] = i;
}

RT_DEV uint32_t owner_of(const PathIO& io, uint32_t j) {
    if (io.fusedOwner != kNotFused) return io.fusedOwner;
    return io.iteration == 0u ? io.pixelList[j] : (uint32_t)__float_as_int(io.raysIn[(size_t)j * io.raysPer * 3].w);
}

// ============================================================ techniques 0-4 (Renderer.cu:565-1284): sample loop x bounce loop
// state: S0 = throughput, seed | S1 = radiance, sample | bounce << 16
// LOCAL (k_path_fused): state, ray and hit of the previous step are the thread's own six quads (PathIO::local) instead of records in memory
template <int TECH, bool LOCAL = false>
RT_DEV bool path_step(const DevScene& sc, const DevCamera& cam, const DevFrame& fr, const DevSettings& st, const PathIO& io, uint32_t j, RayRec& out) {
    const uint32_t i = owner_of(io, j), x = i % fr.W, y = i / fr.W;
    float4* S = LOCAL ? io.local : io.state + (size_t)i * io.stateStride;
    const int nSamples = (TECH == T_BRUTE) ? 1 : (int)st.sampleCount;
    uint32_t seed; int s = 0, b = 0; f3 T = splat3(1.0f), radiance = splat3(0.0f), ro = splat3(0.0f), rd = splat3(0.0f);
    bool open = false;                                                     // a sample's path is in flight and needs its next ray
    if (io.iteration == 0u) seed = i * fr.frameIndex;
    else {
        const float4 s0 = S[0], s1 = S[1];
        T = xyz(s0); seed = (uint32_t)__float_as_int(s0.w); radiance = xyz(s1);
        const uint32_t sb = (uint32_t)__float_as_int(s1.w); s = (int)(sb & 0xFFFFu); b = (int)(sb >> 16);
        const float4* R = LOCAL ? io.local + 2 : io.raysIn + (size_t)j * 3;
        ro = xyz(R[0]); rd = xyz(R[1]);
        const Hit h = LOCAL ? load_hit(io.local, 5u) : load_hit(io.hitsIn, j);
        if (h.tri < 0) radiance = radiance + T * st.sky;
        else {
            const Payload hit = make_hit(sc, ro, rd, h);
            const Mat m = load_mat(sc, tri_material(sc, hit.objectIndex));
            const f3 em = emission(m);
            if (length(em) > 0.0f) radiance = radiance + T * em;
            else {
                const f3 alb = sample_albedo(sc, m, hit.u, hit.v);
                float primaryRoughness = 0.0f;                              // GGX bounces use the PRIMARY hit's roughness (Renderer.cu:1091-1092)
                if (TECH == T_GGX) primaryRoughness = load_mat(sc, tri_material(sc, fr.payload[i].objectIndex)).roughness;
                float bpdf;
                const f3 bdir = sample_dir<TECH>(nrm3(hit), -rd, m, alb, primaryRoughness, seed, bpdf);
                const f3 bbrdf = eval_brdf(nrm3(hit), -rd, bdir, alb, m.metallic, m.roughness);
                const float bcos = gmax(dot(bdir, nrm3(hit)), 0.0f);
                if (TECH == T_COSINE) bpdf = pdf_cosine(bcos);
                T = T * ((bbrdf * bcos) / bpdf);
                ro = pos3(hit) + nrm3(hit) * 1e-12f; rd = bdir;
                ++b;
                open = b < (int)st.maxBounces;
            }
        }
        if (!open) ++s;
    }
    if (!open && s < nSamples) {                                           // start the next sample(s) from the primary hit
        const Payload pp = fr.payload[i];
        const Mat hm = load_mat(sc, tri_material(sc, pp.objectIndex));
        const f3 palbedo = sample_albedo(sc, hm, pp.u, pp.v);
        const f3 pd = ray_direction(cam, x, y);
        for (; s < nSamples; ++s) {
            if (TECH != T_BRUTE) seed += (uint32_t)((s + 1) * 27);
            float pdf;
            const f3 dir = sample_dir<TECH>(nrm3(pp), -pd, hm, palbedo, hm.roughness, seed, pdf);
            const f3 brdf = eval_brdf(nrm3(pp), -pd, dir, palbedo, hm.metallic, hm.roughness);
            const float cosT = gmax(dot(dir, nrm3(pp)), 0.0f);
            if (TECH == T_COSINE) pdf = pdf_cosine(cosT);
            T = splat3(1.0f) * ((brdf * cosT) / pdf);
            ro = pos3(pp) + nrm3(pp) * 1e-12f; rd = dir; b = 0;
            if (st.maxBounces > 0u) { open = true; break; }
        }
    }
    if (open) {
        seed += (uint32_t)(((TECH == T_BRUTE) ? 0 : s) + 31 * b);
        out = ray_closest(ro, rd, i);
        S[0] = f3w(T, seed); S[1] = f3w(radiance, (uint32_t)s | ((uint32_t)b << 16));
        return true;
    }
    if (TECH != T_BRUTE) radiance = radiance / (float)st.sampleCount;
    epilogue(fr, i, rgb1(radiance));
    return false;
}

// ============================================================ LIGHT_SOURCE_SAMPLING (Renderer.cu:1287-1408): one shadow ray per sample
// state: S0 = pending contribution T, seed | S1 = radiance, sample
template <bool LOCAL = false>
RT_DEV bool light_step(const DevScene& sc, const DevCamera& cam, const DevFrame& fr, const DevSettings& st, const PathIO& io, uint32_t j, RayRec& out) {
    const uint32_t i = owner_of(io, j), x = i % fr.W, y = i / fr.W;
    float4* S = LOCAL ? io.local : io.state + (size_t)i * io.stateStride;
    uint32_t seed; int s = 0; f3 radiance = splat3(0.0f);
    if (io.iteration == 0u) seed = i * fr.frameIndex;
    else {
        const float4 s0 = S[0], s1 = S[1];
        const f3 T = xyz(s0); seed = (uint32_t)__float_as_int(s0.w); radiance = xyz(s1); s = __float_as_int(s1.w);
        const uint32_t lightTri = (uint32_t)__float_as_int(LOCAL ? io.local[3].w : io.raysIn[(size_t)j * 3 + 1].w);
        const float4 hq = LOCAL ? io.local[5] : io.hitsIn[j];
        if (hq.x < 0.0f) radiance = radiance + T * st.sky;
        else if ((uint32_t)__float_as_int(hq.w) == lightTri) {
            const Mat lm = load_mat(sc, __float_as_int(sc.triPos[(size_t)lightTri * 3].w));
            if (length(emission(lm)) > 0.0f) radiance = radiance + T * emission(lm);
        }
        ++s;
    }
    if (s < (int)st.sampleCount) {
        const Payload pp = fr.payload[i];
        const Mat hm = load_mat(sc, tri_material(sc, pp.objectIndex));
        const f3 albedo = sample_albedo(sc, hm, pp.u, pp.v);
        const f3 pd = ray_direction(cam, x, y);
        seed += (uint32_t)((s + 1) * 27);
        const PickedLight pl = pick_light(sc, pos3(pp), seed);
        const TriGeom g = load_tri(sc, pl.tri);
        const f3 ep = tri_random_point(g, seed);
        f3 dir = ep - pos3(pp);
        const float dist = length(pos3(pp) - ep);
        dir = dir / dist;
        const f3 brdf = eval_brdf(nrm3(pp), -pd, dir, albedo, hm.metallic, hm.roughness);
        const float cx = gmax(dot(dir, nrm3(pp)), 0.0f);
        const float cy = gmax(dot(-dir, tri_normal(g)), 0.0f);
        const float triAreaPDF = 1.0f / tri_area(g);
        const float totalPDF = (pl.pmf * triAreaPDF) * (dist * dist);
        const f3 T = splat3(1.0f) * (((brdf * cx) * cy) / totalPDF);
        out = ray_shadow(sc, pos3(pp) + nrm3(pp) * 1e-12f, dir, i, pl.tri);
        S[0] = f3w(T, seed); S[1] = f3w(radiance, (uint32_t)s);
        return true;
    }
    radiance = radiance / (float)st.sampleCount;
    epilogue(fr, i, rgb1(radiance));
    return false;
}

// ============================================================ NEE + BRDF MIS (Renderer.cu:1411-1626): per bounce a shadow ray AND the bounce ray
// Three kernels per step, so that each light-tree descent runs on densely packed lanes and no kernel holds two of them:
//   k_shade<T_NEE>  (nee_consume)  results of the previous step's rays: the prepared direct term if the light is visible; bounce ray ->
//                   sky / emitter / surface.  A path that goes on is appended to the PICK list (ballot compaction) with its shading
//                   point in the state record; a path whose BRDF ray hit an emitter to the MIS list (a few percent of the paths —
//                   but nearly every wave holds one, and ComputeDirectEmitterPMF is a whole light-tree descent: inline it ran at
//                   2-5 % lane utilisation in every wave); a path that ended otherwise starts its next sample or is finished.
//   k_nee_mis       one thread per MIS-list path: MIS weight of the BRDF-sampled emitter hit (R.cu:1588-1612), then next sample
//                   (-> PICK list) or the pixel's epilogue.
//   k_nee_emit      one thread per PICK-list path: light pick (light-tree descent), light point, direct term + MIS weight, BRDF sample
//                   -> shadow ray + bounce ray at the entry's own slot (every entry emits: no further compaction).
// state: S0 = throughput, seed | S1 = radiance, sample | bounce << 16 | S2 = pending direct contribution, pdfBRDF
//        S3 = shading point, u | S4 = shading normal, v | S5 = incoming direction, triangle
// next sample of a path that ended (restart from the primary hit), or false when all samples are done
RT_DEV bool nee_next_sample(const DevCamera& cam, const DevFrame& fr, const DevSettings& st, uint32_t i, int s, uint32_t& seed, float4* S, f3 radiance) {
    if (!(s < (int)st.sampleCount && st.maxBounces > 0u)) return false;
    seed += (uint32_t)((s + 1) * 31);
    const Payload pp = fr.payload[i];
    const f3 pd = ray_direction(cam, i % fr.W, i / fr.W);
    S[0] = f3w(splat3(1.0f), seed); S[1] = f3w(radiance, (uint32_t)s); S[2] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    S[3] = make_float4(pp.px, pp.py, pp.pz, pp.u); S[4] = make_float4(pp.nx, pp.ny, pp.nz, pp.v); S[5] = f3w(pd, (uint32_t)pp.objectIndex);
    return true;
}
// returns 1: goes on (PICK list), 2: BRDF ray hit an emitter (MIS list), 0: pixel finished
RT_DEV int nee_consume(const DevScene& sc, const DevCamera& cam, const DevFrame& fr, const DevSettings& st, const PathIO& io, uint32_t j, uint32_t& owner) {
    const uint32_t i = owner_of(io, j);
    owner = i;
    float4* S = io.state + (size_t)i * io.stateStride;
    const uint32_t maxBounces = st.maxBounces;
    uint32_t seed; int s = 0; uint32_t bounce = 0; f3 T = splat3(1.0f), radiance = splat3(0.0f), rd = splat3(0.0f); float pdfBRDF = 1.0f;
    Payload hit;
    bool open = false;
    if (io.iteration == 0u) seed = i * fr.frameIndex;
    else {
        const float4 s0 = S[0], s1 = S[1], s2 = S[2];
        T = xyz(s0); seed = (uint32_t)__float_as_int(s0.w); radiance = xyz(s1); pdfBRDF = s2.w;
        const uint32_t sb = (uint32_t)__float_as_int(s1.w); s = (int)(sb & 0xFFFFu); bounce = sb >> 16;
        const float4* R = io.raysIn + (size_t)j * io.raysPer * 3;
        const uint32_t lightTri = (uint32_t)__float_as_int(R[1].w);
        const float4 sh = io.hitsIn[(size_t)j * io.raysPer];
        if (sh.x > 0.0f && (uint32_t)__float_as_int(sh.w) == lightTri) radiance = radiance + xyz(s2);      // the direct term prepared by k_nee_emit
        if (maxBounces != 1u) {
            const f3 ro = xyz(R[3]); rd = xyz(R[4]);
            const Hit h = load_hit(io.hitsIn, j * 2u + 1u);
            if (h.tri < 0) radiance = radiance + T * st.sky;
            else {
                hit = make_hit(sc, ro, rd, h);
                const Mat em = load_mat(sc, tri_material(sc, hit.objectIndex));
                if (length(emission(em)) > 0.0f) {                         // -> k_nee_mis (with the radiance so far and the hit point)
                    S[0] = f3w(T, seed); S[1] = f3w(radiance, (uint32_t)s | (bounce << 16)); S[2] = make_float4(0.0f, 0.0f, 0.0f, pdfBRDF);
                    S[3] = make_float4(hit.px, hit.py, hit.pz, 0.0f); S[5] = f3w(rd, (uint32_t)hit.objectIndex);
                    return 2;
                } else {
                    ++bounce;
                    open = bounce < maxBounces;
                }
            }
        }
        if (!open) ++s;
    }
    if (open) {
        S[0] = f3w(T, seed); S[1] = f3w(radiance, (uint32_t)s | (bounce << 16)); S[2] = make_float4(0.0f, 0.0f, 0.0f, pdfBRDF);
        S[3] = make_float4(hit.px, hit.py, hit.pz, hit.u); S[4] = make_float4(hit.nx, hit.ny, hit.nz, hit.v); S[5] = f3w(rd, (uint32_t)hit.objectIndex);
        return 1;
    }
    if (nee_next_sample(cam, fr, st, i, s, seed, S, radiance)) return 1;
    epilogue(fr, i, rgb1(radiance / (float)st.sampleCount));
    return 0;
}

// MIS weight of a BRDF-sampled emitter hit (R.cu:1588-1612) for the paths nee_consume listed; the sample ends with it
__global__ __launch_bounds__(kBlock) void k_nee_mis(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st, const uint32_t* misList, const uint32_t* misCount,
                                                    float4* state, uint32_t stateStride, uint32_t* pickList, uint32_t* pickCount) {
    const uint32_t n = *misCount;
    for (uint32_t base = blockIdx.x * (uint32_t)kBlock; base < n; base += gridDim.x * (uint32_t)kBlock) {
        const uint32_t j = base + threadIdx.x;
        bool live = false; uint32_t i = 0;
        if (j < n) {
            i = misList[j];
            float4* S = state + (size_t)i * stateStride;
            const float4 s0 = S[0], s1 = S[1], s3 = S[3], s5 = S[5];
            const f3 T = xyz(s0); uint32_t seed = (uint32_t)__float_as_int(s0.w); f3 radiance = xyz(s1);
            int s = (int)((uint32_t)__float_as_int(s1.w) & 0xFFFFu);
            const float pdfBRDF = S[2].w;
            const f3 hpos = xyz(s3);
            const uint32_t tri = (uint32_t)__float_as_int(s5.w);
            const Mat em = load_mat(sc, tri_material(sc, (int)tri));
            const TriGeom eg = load_tri(sc, tri);
            const f3 lp2 = tri_random_point(eg, seed);
            f3 ld2 = lp2 - hpos;
            const float dist2 = length(ld2);
            ld2 = ld2 / dist2;
            const float cy = gmax(dot(-ld2, tri_normal(eg)), 1e-12f);
            const float triAreaPDF = 1.0f / tri_area(eg);
            const float lsa = (triAreaPDF * (dist2 * dist2)) / cy;
            const float pdfDirect = direct_emitter_pmf(sc, hpos, tri) * lsa;
            const float wB = pdfBRDF / gmax(pdfBRDF + pdfDirect, 1e-12f);
            radiance = radiance + (wB * T) * emission(em);
            ++s;
            live = nee_next_sample(cam, fr, st, i, s, seed, S, radiance);
            if (!live) epilogue(fr, i, rgb1(radiance / (float)st.sampleCount));
        }
        const uint32_t slot = block_append(live, pickCount);
        if (live) pickList[slot] = i;
    }
}

// top of the bounce loop (R.cu:1452-1570) for the paths nee_consume listed
#ifndef RT_NEE_EMIT_WAVES
#define RT_NEE_EMIT_WAVES __attribute__((amdgpu_waves_per_eu(5)))      // (the four flat-axis variants of the cluster importance raise the allocator's choice to 100 VGPRs = 4 waves; 96 fit without a spill)
#endif
__global__ __launch_bounds__(kBlock) RT_NEE_EMIT_WAVES void k_nee_emit(DevScene sc, DevSettings st, const uint32_t* pixelList, const uint32_t* count, float4* state, uint32_t stateStride,
                                                     float4* raysOut, uint32_t raysPer) {
    const uint32_t n = *count;
    for (uint32_t j = blockIdx.x * (uint32_t)kBlock + threadIdx.x; j < n; j += gridDim.x * (uint32_t)kBlock) {
        const uint32_t i = pixelList[j];
        float4* S = state + (size_t)i * stateStride;
        const float4 s0 = S[0], s3 = S[3], s4 = S[4], s5 = S[5];
        f3 T = xyz(s0); uint32_t seed = (uint32_t)__float_as_int(s0.w);
        const f3 hpos = xyz(s3), hnrm = xyz(s4), rd = xyz(s5);
        const int hitTri = __float_as_int(s5.w);
        const uint32_t maxBounces = st.maxBounces;
        const PickedLight pl = pick_light(sc, hpos, seed);
        const Mat mat = load_mat(sc, tri_material(sc, hitTri));
        const f3 albedo = sample_albedo(sc, mat, s3.w, s4.w);
        const TriGeom g = load_tri(sc, pl.tri);
        const f3 lp = tri_random_point(g, seed);
        f3 ld = lp - hpos;
        const float dist = length(ld);
        ld = ld / dist;
        f3 C;
        {
            const f3 brdf = eval_brdf(hnrm, -rd, ld, albedo, mat.metallic, mat.roughness);
            const float cx = gmax(dot(ld, hnrm), 0.0f);
            const float cy = gmax(dot(-ld, tri_normal(g)), 1e-12f);
            const float triAreaPDF = 1.0f / tri_area(g);
            const float lsa = (triAreaPDF * (dist * dist)) / cy;
            const float pdfDirect = pl.pmf * lsa;
            const float pdfB = pdf_brdf(hnrm, -rd, ld, albedo, mat.metallic, mat.roughness);
            const f3 em = emission(load_mat(sc, g.mat));
            if (maxBounces == 1u) C = (((T * brdf) * cx) * em) / pdfDirect;
            else { const float wD = pdfDirect / gmax(pdfB + pdfDirect, 1e-12f); C = ((((wD * T) * brdf) * cx) * em) / pdfDirect; }
        }
        const f3 origin = hpos + hnrm * 1e-12f;
        // the consume step adds C if the light is reached and nothing otherwise: a C of exactly zero (the surface faces away from the picked
        // light, a black emitter) makes the shadow ray's answer irrelevant — its slot then holds no ray (kRayNone, DevSettings::skipDeadRays)
        if (st.skipDeadRays && zero3(C)) { RayRec none = ray_closest(origin, ld, i); none.q1.w = __int_as_float((int)kRayNone); store_ray(raysOut, j * raysPer, none); }
        else store_ray(raysOut, j * raysPer, ray_shadow(sc, origin, ld, i, pl.tri));
        float pdfBRDF = S[2].w;
        if (maxBounces != 1u) {
            const f3 nd = sample_brdf(hnrm, -rd, albedo, mat.metallic, mat.roughness, seed, pdfBRDF);
            pdfBRDF = gmax(pdfBRDF, 1e-12f);
            const f3 brdf = eval_brdf(hnrm, -rd, nd, albedo, mat.metallic, mat.roughness);
            const float cosT = dot(nd, hnrm);
            T = T * ((brdf * cosT) / pdfBRDF);
            store_ray(raysOut, j * 2u + 1u, ray_closest(origin, nd, i));
        }
        S[0] = f3w(T, seed); S[2] = f3f(C, pdfBRDF);
    }
}

// ============================================================ ReSTIR GI Part 1 (Renderer.cu:2043-2293)
// primary rays over the band + halo rows; finished pixels get an empty reservoir, the others go on to the bounce loop
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_gi_primary(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st, uint32_t p1Begin, uint32_t p1End, uint32_t extraRow,
                                                       uint32_t* pixelList, uint32_t* listCount) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch
    uint32_t x, y;
    const bool inside = p1_pixel_of_thread(fr, p1Begin, p1End, extraRow, x, y);
    const uint32_t i = x + y * fr.W;
    bool live = false;
    if (inside) {
        const bool inBand = (y >= fr.rowBegin && y < fr.rowEnd);
        const f3 pd = ray_direction(cam, x, y);
        const Payload pp = trace_ray<COUNT>(sc, cam.position, pd, s_stack + threadIdx.x);
        fr.payload[i] = pp;
        const f2 ncur = oct_encode(nrm3(pp));
        fr.normalCur[i] = ncur;
        bool finished = false; f3 finalColor = splat3(0.0f);
        if (pp.hitDistance < 0.0f) { finished = true; finalColor = st.sky; }
        else {
            const Mat hm = load_mat(sc, tri_material(sc, pp.objectIndex));
            if (length(emission(hm)) > 0.0f) { finished = true; finalColor = emission(hm); }
        }
        if (finished) {
            GIRes R; gi_reset(R);
            fr.gi[i] = R; fr.depth[i] = pp.hitDistance;
            fr.giHot[(size_t)i * 4] = make_float4(pp.hitDistance, ncur.x, ncur.y, 0.0f);       // an empty reservoir: |Lo| = 0 (the rest of the record is never read then)
            if (inBand) epilogue(fr, i, rgb1(finalColor));
        } else live = true;
    }
    const uint32_t slot = block_append(live, listCount);
    if (live) pixelList[slot] = i;
}

// state: S0 = throughput, seed | S1 = Lo, bounce | S2 = sample point | S3 = sample normal   (S2 / S3 are written by the step that sees the
// first bounce's hit and read by the step that finishes the path only: the steps are HBM-bound, every quad not moved counts)
// returns true while the bounce path needs another ray; `toPart2` = the pixel's reservoir is final and the pixel is inside the band
RT_DEV bool gi1_step(const DevScene& sc, const DevCamera& cam, const DevFrame& fr, const DevSettings& st, const PathIO& io, uint32_t j, RayRec& out, bool& toPart2) {
    const uint32_t i = owner_of(io, j), x = i % fr.W, y = i / fr.W;
    float4* S = io.state + (size_t)i * io.stateStride;
    uint32_t seed; int b = 0;
    f3 T = splat3(1.0f), Lo = splat3(0.0f), samplePoint = splat3(0.0f), sampleNormal = splat3(0.0f), ro, rd;
    bool open, haveSample = true;                                          // haveSample: samplePoint / sampleNormal are in registers (not only in S2 / S3)
    if (io.iteration == 0u) {
        const Payload pp = fr.payload[i];
        seed = i * (fr.frameIndex + 1u + st.randSeed);
        const f3 pd = ray_direction(cam, x, y);
        const Mat hm = load_mat(sc, tri_material(sc, pp.objectIndex));
        const f3 albedo = sample_albedo(sc, hm, pp.u, pp.v);
        float pdf;
        const f3 dir = sample_brdf(nrm3(pp), -pd, albedo, hm.metallic, hm.roughness, seed, pdf);
        const f3 brdf = eval_brdf(nrm3(pp), -pd, dir, albedo, hm.metallic, hm.roughness);
        const float cosT = gmax(dot(dir, nrm3(pp)), 0.0f);
        T = T * ((brdf * cosT) / pdf);
        ro = pos3(pp) + nrm3(pp) * 1e-12f; rd = dir;
        open = st.maxBounces > 0u;
    } else {
        const float4 s0 = S[0], s1 = S[1];
        T = xyz(s0); seed = (uint32_t)__float_as_int(s0.w); Lo = xyz(s1); b = __float_as_int(s1.w);
        const float4* R = io.raysIn + (size_t)j * 3;
        ro = xyz(R[0]); rd = xyz(R[1]);
        const Hit h = load_hit(io.hitsIn, j);
        const Payload hit = (h.tri < 0) ? make_miss() : make_hit(sc, ro, rd, h);
        if (b == 0) { samplePoint = pos3(hit); sampleNormal = nrm3(hit); }
        else haveSample = false;
        open = false;
        if (hit.hitDistance < 0.0f) Lo = Lo + T * st.sky;
        else {
            const Mat m = load_mat(sc, tri_material(sc, hit.objectIndex));
            const f3 em = emission(m);
            if (length(em) > 0.0f) Lo = Lo + T * em;
            else {
                const f3 alb = sample_albedo(sc, m, hit.u, hit.v);
                float bpdf;
                const f3 bdir = sample_brdf(nrm3(hit), -rd, alb, m.metallic, m.roughness, seed, bpdf);
                const f3 bbrdf = eval_brdf(nrm3(hit), -rd, bdir, alb, m.metallic, m.roughness);
                const float bcos = gmax(dot(bdir, nrm3(hit)), 0.0f);
                T = T * ((bbrdf * bcos) / bpdf);
                ro = pos3(hit) + nrm3(hit) * 1e-12f; rd = bdir;
                ++b;
                open = b < (int)st.maxBounces;
            }
        }
    }
    toPart2 = false;
    if (open) {
        if (io.iteration != 0u && b == 1) { S[2] = f3f(samplePoint, 0.0f); S[3] = f3f(sampleNormal, 0.0f); }      // (b was 0 when this step started)
        seed += (uint32_t)(31 * b);
        out = ray_closest(ro, rd, i);
        S[0] = f3w(T, seed); S[1] = f3w(Lo, (uint32_t)b);
        return true;
    }
    // the path is complete: initial reservoir (R.cu:2186-2228), temporal reuse (:2230-2290)
    if (!haveSample) { samplePoint = xyz(S[2]); sampleNormal = xyz(S[3]); }
    const Payload pp = fr.payload[i];
    const uint32_t originalSeed = i * (fr.frameIndex + 1u + st.randSeed);  // the seed the path started from (R.cu:2094)
    GIRes R; gi_reset(R);
    {
        GISample sm; sm.seed = originalSeed;
        sm.vp[0] = pp.px; sm.vp[1] = pp.py; sm.vp[2] = pp.pz;
        const f2 vn = oct_encode(nrm3(pp)); sm.vn[0] = vn.x; sm.vn[1] = vn.y;
        sm.sp[0] = samplePoint.x; sm.sp[1] = samplePoint.y; sm.sp[2] = samplePoint.z;
        const f2 sn = oct_encode(sampleNormal); sm.sn[0] = sn.x; sm.sn[1] = sn.y;
        sm.Lo[0] = Lo.x; sm.Lo[1] = Lo.y; sm.Lo[2] = Lo.z; sm.pdf = 0.0f;
        const float len = length(Lo);
        gi_update(R, sm, len, 1u, len, seed);
        R.W = R.s.pdf > 0.0f ? ((1.0f / R.s.pdf) * R.wSum) / (float)R.M : 0.0f;
    }
    if (st.useTemporal) {
        uint32_t prow;
        const uint32_t prevIdx = prev_pixel(cam, pos3(pp), prow);
        const f3 prevN = oct_decode(fr.normalPrev[prevIdx]);
        GIRes prev = fr.giPrev[prevIdx];
        const bool valid = (double)dot(prevN, nrm3(pp)) >= 0.99 && prow >= fr.histBegin && prow < fr.histEnd;
        const f3 plo = lo3(prev.s);
        if (valid && prev.M > 0u && dot(plo, plo) > 0.0f) {
            GIRes Tm = R;
            const uint32_t lim = st.historyLimit * R.M;
            prev.M = (lim < prev.M) ? lim : prev.M;
            const float pdf = length(plo);
            gi_update(Tm, prev.s, (pdf * prev.W) * (float)prev.M, prev.M, pdf, seed);
            Tm.W = Tm.s.pdf > 0.0f ? Tm.s.pdf / ((float)Tm.M * Tm.s.pdf) : 0.0f;
            gi_reset(R);
            gi_merge(R, Tm, Tm.s.pdf, seed);
        }
    }
    fr.gi[i] = R;
    {   // the neighbour record of Part 2 (DevFrame::giHot): payload.hitDistance, normalCur, |Lo| as Part 2 would compute them + what a merge reads
        const f2 n = oct_encode(nrm3(pp));
        float4* q = fr.giHot + (size_t)i * 4;
        q[0] = make_float4(pp.hitDistance, n.x, n.y, length(lo3(R.s)));
        q[1] = make_float4(R.s.vp[0], R.s.vp[1], R.s.vp[2], __int_as_float((int)R.M));
        q[2] = make_float4(R.s.sp[0], R.s.sp[1], R.s.sp[2], R.wSum);
        q[3] = make_float4(R.s.sn[0], R.s.sn[1], 0.0f, 0.0f);
    }
    toPart2 = (y >= fr.rowBegin && y < fr.rowEnd);          // replaces the reference's image sentinel (R.cu:2746-2750 / :2787): Part 2 runs on a list
    return false;
}

// ============================================================ ReSTIR GI Part 2 (Renderer.cu:2295-2387): one visibility ray per accepted neighbour
// The steps are HBM-bound (0.56 L2 hit rate, 5.7 TB/s), so the state between two steps is as small as the algorithm allows: the
// reservoir's sample is always SOME pixel's Part-1 sample (its own, or the neighbour's that a merge selected) with only `pdf` replaced
// (ReSTIR_GI_Reservoir.cu:5-34), and Part 1's reservoirs do not change during Part 2 — so the state names that pixel instead of carrying
// the 60-byte sample; and a merge needs nothing of the neighbour but its weightSum and M, which are kept from the step that emitted the
// visibility ray (no second gather of the neighbour's reservoir).
// state: S0 = sample's source pixel, sample pdf, W, weightSum | S1 = M, seed, neighbour counter, Z | S2 = pending neighbour: pixel, pdf, its weightSum, its M
//        S3 = the current sample's visible point (the only part of the sample a step needs) | S4 = the pending neighbour's visible point
RT_DEV bool gi2_step(const DevScene& sc, const DevCamera& cam, const DevFrame& fr, const DevSettings& st, const PathIO& io, uint32_t j, RayRec& out) {
    const uint32_t i = io.ownersIn[j], x = i % fr.W, y = i / fr.W;
    float4* S = io.state + (size_t)i * io.stateStride;
    const Payload pp = fr.payload[i];
    uint32_t src, seed, n = 0, Z = 0, M; float spdf, Wr, wSum; f3 rvp; bool rvpChanged = true;
    if (io.iteration == 0u) {
        const GIRes own = fr.gi[i];
        src = i; spdf = own.s.pdf; Wr = own.W; wSum = own.wSum; M = own.M; rvp = mk3(own.s.vp[0], own.s.vp[1], own.s.vp[2]);
        seed = i * (fr.frameIndex + 213u + st.randSeed);
        if (st.useSpatial) { const float plen = length(lo3(own.s)); Z = plen > 0.0f ? M : 0u; }
    } else {
        const float4 s0 = S[0], s1 = S[1], s2 = S[2];
        src = (uint32_t)__float_as_int(s0.x); spdf = s0.y; Wr = s0.z; wSum = s0.w;
        M = (uint32_t)__float_as_int(s1.x); seed = (uint32_t)__float_as_int(s1.y); n = (uint32_t)__float_as_int(s1.z); Z = (uint32_t)__float_as_int(s1.w);
        const uint32_t ni = (uint32_t)__float_as_int(s2.x);
        float pdf = s2.y;
        const float nWSum = s2.z; const uint32_t nM = (uint32_t)__float_as_int(s2.w);
        rvp = xyz(S[3]); rvpChanged = false;
        if (!(io.hitsIn[j].x != 0.0f)) pdf = 0.0f;                          // R.cu:2356-2366: the neighbour's sample point is not visible
        // gi_merge(R, N, pdf, seed) on the counters (ReSTIR_GI_Reservoir.cu:36-43 -> :5-34)
        const uint32_t prevM = M;
        const float w = (pdf * nWSum) * (float)nM;
        wSum += w; M += 1u;
        if (rnd(seed) < w / wSum) { src = ni; spdf = pdf; rvp = xyz(S[4]); rvpChanged = true; }
        M = prevM + nM;
        ++n;
    }
    if (st.useSpatial) {
        for (; n < st.numNeighbors; ++n) {
            const uint32_t ni = neighbor_index(cam, fr.W, x, y, st.radius, seed);
            const float4* nq = fr.giHot + (size_t)ni * 4;
            const float4 hot = nq[0];                                        // depth, normal, |Lo| of the neighbour: one 16-byte gather decides
            const float nd = hot.x, pdp = pp.hitDistance, nlen = hot.w;
            f2 nnrm; nnrm.x = hot.y; nnrm.y = hot.z;
            if ((nd > 1.1f * pdp || nd < 0.9f * pdp) || (double)dot(nrm3(pp), oct_decode(nnrm)) < 0.906 || nlen == 0.0f) continue;
            const float4 n1 = nq[1], n2 = nq[2], n3 = nq[3];                 // accepted: the rest of the SAME 64-byte line (not the 80-byte reservoir, two or three lines away)
            const uint32_t NM = (uint32_t)__float_as_int(n1.w); const float NwSum = n2.w;
            Z += NM;
            f2 sne; sne.x = n3.x; sne.y = n3.y;
            const f3 sn = oct_decode(sne);
            const f3 nvp = xyz(n1), nsp = xyz(n2);
            const f3 dQ = normalize(nvp - nsp);
            const float cosQ = dot(sn, dQ);
            const f3 dR = normalize(rvp - nsp);
            const float cosR = dot(sn, dR);
            const float jl = cosQ > 0.0f ? cosR / cosQ : 0.0f;
            const float distQ = length(nvp - nsp), distR = length(rvp - nsp);
            const float jr = distR > 0.0f ? (distQ * distQ) / (distR * distR) : 0.0f;
            const float jac = jl * jr;
            const float pdf = jac > 0.0f ? nlen / jac : 0.0f;
            const float tol = gmax(1e-4f, distR * 1e-3f);
            out = ray_visible(nsp, dR, i, distR, tol);
            if (st.skipDeadRays && pdf == 0.0f) out.q1.w = __int_as_float((int)kRayNone);     // a merge weight of zero whatever the ray finds (R.cu:2356-2368): no ray in this slot
            S[0] = make_float4(__int_as_float((int)src), spdf, Wr, wSum);
            S[1] = make_float4(__int_as_float((int)M), __int_as_float((int)seed), __int_as_float((int)n), __int_as_float((int)Z));
            S[2] = make_float4(__int_as_float((int)ni), pdf, NwSum, __int_as_float((int)NM));
            if (rvpChanged) S[3] = f3f(rvp, 0.0f);
            S[4] = f3f(nvp, 0.0f);
            return true;
        }
    }
    GIRes R = fr.gi[src];                                                   // the selected sample, with the reservoir's own pdf / counters
    R.s.pdf = spdf; R.M = M; R.wSum = wSum; R.W = Wr;
    if (st.useSpatial) R.W = R.s.pdf > 0.0f ? R.s.pdf / ((float)Z * R.s.pdf) : 0.0f;
    const f3 radiance = lo3(R.s) * R.W;
    fr.depth[i] = pp.hitDistance;
    fr.giPrev[i] = R;
    epilogue(fr, i, rgb1(radiance));
    return false;
}

// ============================================================ ReSTIR GI Part 2 in ONE launch (tuning key 19)
// The staged Part 2 is 2 x neighbours + 1 launches whose shade steps are HBM-bound: every step moves the pixel's state (64-80 B each way), a
// 48-byte ray record and a 16-byte result through memory, and reads the payload again.  Here one thread keeps its pixel for the whole
// neighbour loop — same expressions, same random draws in the same order as gi2_step — with the visibility ray traced in place
// (trace_one, the one-thread-per-ray traversal of the small-scene path): the state never leaves registers, nothing but the payload, the
// reservoirs and the neighbours' hot records is read, nothing but the final reservoir, depth and the pixel is written.  The wave walks in
// rounds — every lane looks for its next accepted neighbour, then the lanes that found one trace together — so a ray's node loop runs with
// all the lanes that still have a ray; a lane whose neighbours are used up idles until the wave's last lane is done (what the staged
// path's compaction avoids, at the price of the traffic above).
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_gi2_fused(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st, const uint32_t* list, const uint32_t* listCount) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch
    const uint32_t count = *listCount;
    for (uint32_t base = blockIdx.x * (uint32_t)kBlock; base < count; base += gridDim.x * (uint32_t)kBlock) {   // (block-uniform trips: node_step's ballots see whole waves)
        const uint32_t j = base + threadIdx.x;
        const bool inside = j < count;
        const uint32_t i = inside ? list[j] : 0u, x = i % fr.W, y = i / fr.W;
        uint32_t src = i, seed = 0, n = 0, Z = 0, M = 0; float spdf = 0.0f, Wr = 0.0f, wSum = 0.0f, pdp = 0.0f; f3 rvp = splat3(0.0f), pn = splat3(0.0f);
        if (inside) {
            const Payload pp = fr.payload[i];
            pdp = pp.hitDistance; pn = nrm3(pp);
            const GIRes own = fr.gi[i];
            spdf = own.s.pdf; Wr = own.W; wSum = own.wSum; M = own.M; rvp = mk3(own.s.vp[0], own.s.vp[1], own.s.vp[2]);
            seed = i * (fr.frameIndex + 213u + st.randSeed);
            if (st.useSpatial) { const float plen = length(lo3(own.s)); Z = plen > 0.0f ? M : 0u; }
        }
        while (true) {
            bool have = false; uint32_t ni = 0, NM = 0; float pdf = 0.0f, NwSum = 0.0f, distR = 0.0f; f3 nvp = splat3(0.0f), nsp = splat3(0.0f), dR = splat3(0.0f);
            if (inside && st.useSpatial) {
                for (; n < st.numNeighbors; ++n) {                                   // gi2_step's search for the next accepted neighbour
                    ni = neighbor_index(cam, fr.W, x, y, st.radius, seed);
                    const float4* nq = fr.giHot + (size_t)ni * 4;
                    const float4 hot = nq[0];
                    const float nd = hot.x, nlen = hot.w;
                    f2 nnrm; nnrm.x = hot.y; nnrm.y = hot.z;
                    if ((nd > 1.1f * pdp || nd < 0.9f * pdp) || (double)dot(pn, oct_decode(nnrm)) < 0.906 || nlen == 0.0f) continue;
                    const float4 n1 = nq[1], n2 = nq[2], n3 = nq[3];
                    NM = (uint32_t)__float_as_int(n1.w); NwSum = n2.w;
                    Z += NM;
                    f2 sne; sne.x = n3.x; sne.y = n3.y;
                    const f3 sn = oct_decode(sne);
                    nvp = xyz(n1); nsp = xyz(n2);
                    const f3 dQ = normalize(nvp - nsp);
                    const float cosQ = dot(sn, dQ);
                    dR = normalize(rvp - nsp);
                    const float cosR = dot(sn, dR);
                    const float jl = cosQ > 0.0f ? cosR / cosQ : 0.0f;
                    const float distQ = length(nvp - nsp); distR = length(rvp - nsp);
                    const float jr = distR > 0.0f ? (distQ * distQ) / (distR * distR) : 0.0f;
                    const float jac = jl * jr;
                    pdf = jac > 0.0f ? nlen / jac : 0.0f;
                    have = true;
                    break;
                }
            }
            if (__ballot(have) == 0ull) break;                                       // no lane of the wave has a neighbour left
            if (have) {
                // the visibility ray (R.cu:2356-2366); a merge weight of zero whatever the ray finds is not traced (DevSettings::skipDeadRays)
                bool visible = true;
                if (!(st.skipDeadRays && pdf == 0.0f)) {
                    const float tol = gmax(1e-4f, distR * 1e-3f);
                    visible = trace_one<COUNT>(sc, nsp, dR, kRayVisible, distR, tol, s_stack + threadIdx.x).x != 0.0f;
                }
                if (!visible) pdf = 0.0f;
                // gi_merge(R, N, pdf, seed) on the counters (ReSTIR_GI_Reservoir.cu:36-43 -> :5-34)
                const uint32_t prevM = M;
                const float w = (pdf * NwSum) * (float)NM;
                wSum += w; M += 1u;
                if (rnd(seed) < w / wSum) { src = ni; spdf = pdf; rvp = nvp; }
                M = prevM + NM;
                ++n;
            }
        }
        if (inside) {
            GIRes R = fr.gi[src];                                                   // the selected sample, with the reservoir's own pdf / counters
            R.s.pdf = spdf; R.M = M; R.wSum = wSum; R.W = Wr;
            if (st.useSpatial) R.W = R.s.pdf > 0.0f ? R.s.pdf / ((float)Z * R.s.pdf) : 0.0f;
            const f3 radiance = lo3(R.s) * R.W;
            fr.depth[i] = pdp;
            fr.giPrev[i] = R;
            epilogue(fr, i, rgb1(radiance));
        }
    }
}

// ============================================================ ReSTIR GI Part 2 as ONE PERSISTENT launch (tuning key 19 = 2)
// k_gi2_fused above pays for its simplicity with the traversal: a wave waits for the slowest of its 64 visibility rays five times over and
// a pixel whose neighbours are used up idles until the wave is done (measured: config 5 13.9 ms against the stages' 12.9).  This kernel
// keeps the state in registers too, but runs like the persistent trace kernels: a lane owns one PIXEL of the Part-2 list; whenever
// `refillLanes` lanes of the wave have no ray in flight the wave services them together — merge the finished ray's verdict, look for the
// next accepted neighbour, start its visibility ray, or write the pixel out and take the next one from the queue — and then goes on
// traversing with all the rays it has.  Same expressions and random draws per pixel as gi2_step, in the same order.
struct GI2Queue { const uint32_t* list; const uint32_t* count; uint32_t* head; uint32_t chunk, refillLanes, staticChunks, minChunk; };
#ifndef RT_GI2_WAVES
#define RT_GI2_WAVES __attribute__((amdgpu_waves_per_eu(5)))
#endif
template <bool COUNT>
__global__ __launch_bounds__(kBlock) RT_GI2_WAVES void k_gi2_persistent(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st, GI2Queue q) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch
    int32_t* lds = s_stack + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = *q.count;
    const uint32_t nWaves = gridDim.x * (uint32_t)(kBlock / 64), myWave = blockIdx.x * (uint32_t)(kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t share = (total + nWaves - 1u) / nWaves;
    const uint32_t chunk = q.chunk, want1 = chunk * q.staticChunks;
    const uint32_t first = share < want1 ? (share < 16u ? 16u : share) : want1;
    const uint32_t dynBase = nWaves * first;
    const bool hasDyn = dynBase < total;
    bool more = total != 0u;
    uint32_t chunkNext = myWave * first < total ? myWave * first : total;
    uint32_t chunkEnd = (myWave + 1u) * first < total ? (myWave + 1u) * first : total;
    // the lane's pixel
    bool owns = false, tracing = false, verdict = false, visible = false;
    uint32_t i = 0, src = 0, seed = 0, n = 0, Z = 0, M = 0, ni = 0, NM = 0;
    float spdf = 0.0f, Wr = 0.0f, wSum = 0.0f, pdp = 0.0f, pdf = 0.0f, NwSum = 0.0f; f3 rvp = splat3(0.0f), pn = splat3(0.0f);
    // the lane's ray
    f3 o = splat3(0.0f), d = splat3(0.0f); RayPk pk = make_raypk(o, 1.0f, 1.0f, 1.0f);
    float tL = 0.0f, cut = 0.0f, hu = 0.0f, hv = 0.0f; int32_t cur = kExit; int top = 0;
    uint32_t nBox = 0, nTri = 0, nNode = 0;
    while (true) {
        const unsigned long long serviceable = __ballot(!tracing && (owns || more));
        if (serviceable != 0ull && ((uint32_t)__popcll(serviceable) >= q.refillLanes || __ballot(tracing) == 0ull)) {
            if (owns && verdict) {                                                   // the ray that just ended: gi_merge on the counters (gi2_step)
                if (!visible) pdf = 0.0f;
                const uint32_t prevM = M;
                const float w = (pdf * NwSum) * (float)NM;
                wSum += w; M += 1u;
                if (rnd(seed) < w / wSum) { src = ni; spdf = pdf; rvp = xyz(fr.giHot[(size_t)ni * 4 + 1]); }     // the neighbour's visible point (the staged path parks it in its state)
                M = prevM + NM;
                ++n; verdict = false;
            }
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {                                                     // lanes without a pixel take the next ones of the list
                    const unsigned long long need = __ballot(!owns);
                    if (more && need != 0ull) {
                        if (chunkNext >= chunkEnd && hasDyn) {
                            uint32_t base = 0;
                            uint32_t size = (total - chunkEnd) / nWaves;
                            size = size < q.minChunk ? q.minChunk : (size > chunk ? chunk : size);
                            if (lane == 0u) base = dynBase + atomicAdd(q.head, size);
                            base = (uint32_t)__shfl((int)base, 0);
                            chunkNext = base < total ? base : total;
                            chunkEnd = (base + size < total) ? base + size : total;
                        }
                        const uint32_t slot = chunkNext + (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
                        const uint32_t want = (uint32_t)__popcll(need), avail = chunkEnd - chunkNext;
                        chunkNext += (want < avail) ? want : avail;
                        more = chunkNext < chunkEnd || (hasDyn && chunkEnd < total);
                        if (!owns && slot < chunkEnd) {
                            i = q.list[slot];
                            const Payload pp = fr.payload[i];
                            pdp = pp.hitDistance; pn = nrm3(pp);
                            const GIRes own = fr.gi[i];
                            src = i; spdf = own.s.pdf; Wr = own.W; wSum = own.wSum; M = own.M; rvp = mk3(own.s.vp[0], own.s.vp[1], own.s.vp[2]);
                            seed = i * (fr.frameIndex + 213u + st.randSeed);
                            n = 0; Z = 0;
                            if (st.useSpatial) { const float plen = length(lo3(own.s)); Z = plen > 0.0f ? M : 0u; }
                            owns = true; verdict = false;
                        }
                    }
                }
                if (owns && !tracing) {
                    bool found = false;
                    const uint32_t x = i % fr.W, y = i / fr.W;
                    if (st.useSpatial) {
                        for (; n < st.numNeighbors; ++n) {                           // gi2_step's search for the next accepted neighbour
                            ni = neighbor_index(cam, fr.W, x, y, st.radius, seed);
                            const float4* nq = fr.giHot + (size_t)ni * 4;
                            const float4 hot = nq[0];
                            const float nd = hot.x, nlen = hot.w;
                            f2 nnrm; nnrm.x = hot.y; nnrm.y = hot.z;
                            if ((nd > 1.1f * pdp || nd < 0.9f * pdp) || (double)dot(pn, oct_decode(nnrm)) < 0.906 || nlen == 0.0f) continue;
                            const float4 n1 = nq[1], n2 = nq[2], n3 = nq[3];
                            NM = (uint32_t)__float_as_int(n1.w); NwSum = n2.w;
                            Z += NM;
                            f2 sne; sne.x = n3.x; sne.y = n3.y;
                            const f3 sn = oct_decode(sne);
                            const f3 nvp = xyz(n1), nsp = xyz(n2);
                            const f3 dQ = normalize(nvp - nsp);
                            const float cosQ = dot(sn, dQ);
                            const f3 dR = normalize(rvp - nsp);
                            const float cosR = dot(sn, dR);
                            const float jl = cosQ > 0.0f ? cosR / cosQ : 0.0f;
                            const float distQ = length(nvp - nsp), distR = length(rvp - nsp);
                            const float jr = distR > 0.0f ? (distQ * distQ) / (distR * distR) : 0.0f;
                            const float jac = jl * jr;
                            pdf = jac > 0.0f ? nlen / jac : 0.0f;
                            if (st.skipDeadRays && pdf == 0.0f) {                    // a merge weight of zero whatever the ray finds: merged at once, no ray (DevSettings::skipDeadRays)
                                const uint32_t prevM = M;
                                const float w = (pdf * NwSum) * (float)NM;
                                wSum += w; M += 1u;
                                if (rnd(seed) < w / wSum) { src = ni; spdf = pdf; rvp = nvp; }
                                M = prevM + NM;
                                continue;
                            }
                            const float tol = gmax(1e-4f, distR * 1e-3f);
                            o = nsp; d = dR;
                            pk = make_raypk(o, safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
                            hu = distR - tol; tL = distR + tol; hv = 0.0f;           // hu = dist - tol, interval end = dist + tol; hv: 0 nothing yet, 1 found, -1 blocked
                            cut = tL * 1.000001f;
                            if (COUNT) { nBox = 0; nNode = 0; nTri = 0; }
                            top = 0; lane_push(lds, top, kExit);
                            cur = (sc.triCount == 0 || ray_not_finite(o, d)) ? kExit : sc.rootRef;
                            found = true;
                            break;
                        }
                    }
                    if (found) tracing = true;
                    else {
                        GIRes R = fr.gi[src];                                       // the selected sample, with the reservoir's own pdf / counters
                        R.s.pdf = spdf; R.M = M; R.wSum = wSum; R.W = Wr;
                        if (st.useSpatial) R.W = R.s.pdf > 0.0f ? R.s.pdf / ((float)Z * R.s.pdf) : 0.0f;
                        const f3 radiance = lo3(R.s) * R.W;
                        fr.depth[i] = pdp;
                        fr.giPrev[i] = R;
                        epilogue(fr, i, rgb1(radiance));
                        owns = false;
                    }
                }
            }
        }
        if (__ballot(tracing) == 0ull) { if (!more && __ballot(owns) == 0ull) break; else continue; }
        while (true) {
            bool walk = tracing && cur >= 0;
            const uint32_t quorum = quorum_of(sc.nodeQuorum, (uint32_t)__popcll(__ballot(tracing)));
            while (walk) {
                { Stack stk; stk.lds = lds; stk.top = top; cur = node_step<COUNT>(sc.nodes, sc.stackBudget, cur, pk, cut, stk, nBox, nNode); top = stk.top; }
                walk = cur >= 0;
                if ((uint32_t)__popcll(__ballot(walk)) < quorum) break;
            }
            if (tracing && cur < 0 && cur != kExit) {
                const uint32_t code = (uint32_t)~cur, firstTri = code >> 2, cnt = (code & 3u) + 1u;
                bool done = false;
                for (uint32_t k = 0; k < cnt; ++k) {
                    const float4* tp = sc.leafTris + (size_t)(firstTri + k) * 3;
                    float t, u, v; uint32_t id;
                    if (COUNT) nTri += 1;
                    if (!tri_test(tp, o, d, t, u, v, id)) continue;
                    if (t < hu) { hv = -1.0f; done = true; break; }
                    if (t <= tL) hv = 1.0f;
                }
                cur = done ? kExit : lane_pop(lds, top);
            }
            if (tracing && cur == kExit) {
                visible = hv > 0.0f;
                if (COUNT) {
                    atomicAdd(sc.rayCounter + 0, 1ull); atomicAdd(sc.rayCounter + 1, (unsigned long long)nBox);
                    atomicAdd(sc.rayCounter + 2, (unsigned long long)nTri); atomicAdd(sc.rayCounter + 3, (unsigned long long)(visible ? 1 : 0));
                    atomicAdd(sc.rayCounter + 4, (unsigned long long)nNode);
                }
                tracing = false; verdict = true;
            }
            const unsigned long long act = __ballot(tracing);
            if (act == 0ull) break;
            if ((uint32_t)__popcll(__ballot(!tracing && (owns || more))) >= q.refillLanes) break;
        }
    }
}

// ============================================================ small scenes: the whole frame of techniques 0-5 in ONE launch
// A scene whose rays cost a handful of node visits (Cornell box: 3; the reference's banana: 8) gains nothing from re-packing lanes
// between bounces, and every one of the 2 x steps + 1 stage launches pays its own fill, drain and one-wave-round latency: the
// Cornell frame of BASELINE config 1 took 0.21 ms in ten launches against 0.11 ms for one thread per pixel.  This
// kernel is that one thread per pixel again — primary ray, then the SAME step functions (path_step / light_step: same expressions, same
// random draws, same order of additions) with the same one-thread-per-ray traversal (trace_one) in between — so its pixels are the stage
// path's bit for bit (tests/test_gpu_tuning.py).  The step functions are instantiated with LOCAL = true: what a stage keeps in the state,
// ray and hit records between two steps is the thread's own six quads here (PathIO::local — registers; the first version wrote and re-read the
// records at the thread's own slot, 8 % of the Cornell frame).  Chosen by the host for trees of fewer than 64 k triangles (tuning key 17).
template <int TECH, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_path_fused(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st, PathIO io, uint32_t steps) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch
    uint32_t x, y;
    bool inside = pixel_of_thread(fr, fr.rowBegin, fr.rowEnd, x, y);
    const uint32_t j = (y - fr.rowBegin) * fr.W + x;             // the thread's slot in the ray / hit records: its pixel's index inside the band (local rows of a striped frame)
    if (fr.stripeRows != 0u) {
        const uint32_t k = y / fr.stripeRows;
        y = (k * fr.stripeParts + fr.stripePart) * fr.stripeRows + (y - k * fr.stripeRows);
        inside = inside && y < fr.H;
    }
    const uint32_t i = x + y * fr.W;
    bool live = false;
    if (inside) {
        const f3 pd = ray_direction(cam, x, y);
        const Payload pp = trace_ray<COUNT>(sc, cam.position, pd, s_stack + threadIdx.x);
        fr.payload[i] = pp;
        if (pp.hitDistance < 0.0f) epilogue(fr, i, rgb1(st.sky));
        else {
            const Mat hm = load_mat(sc, tri_material(sc, pp.objectIndex));
            if (length(emission(hm)) > 0.0f) epilogue(fr, i, rgb1(emission(hm))); else live = true;
        }
    }
    io.fusedOwner = i;
    float4 own[6];                                               // the thread's own state (2), ray (3), hit (1): PathIO::local — constant indices, so registers
    io.local = own;
    for (uint32_t it = 0; live && it <= steps; ++it) {
        io.iteration = it;
        RayRec r;
        live = (TECH == T_LIGHT) ? light_step<true>(sc, cam, fr, st, io, j, r) : path_step<(TECH <= T_BRDF ? TECH : T_BRUTE), true>(sc, cam, fr, st, io, j, r);
        if (!live) break;
        own[2] = r.q0; own[3] = r.q1; own[4] = r.q2;
        own[5] = trace_one<COUNT>(sc, xyz(r.q0), xyz(r.q1), (uint32_t)__float_as_int(r.q1.w), r.q2.x, r.q2.y, s_stack + threadIdx.x);
    }
}

// ============================================================ the shade kernel: one thread per live path, grid-stride over the list
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES
#endif
// (register budgets of the shade kernels are the compiler's own — 81 to 121 VGPRs, none spills; forcing 5 waves per SIMD spills in all
// but three of them: tools/kernel_resources.py -DRT_SHADE_WAVES=...)
template <int TECH>
__global__ __launch_bounds__(kBlock) RT_SHADE_WAVES void k_shade(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st, PathIO io) {
    const uint32_t count = *io.countIn;
    for (uint32_t base = blockIdx.x * (uint32_t)kBlock; base < count; base += gridDim.x * (uint32_t)kBlock) {
        const uint32_t j = base + threadIdx.x;
        bool live = false, toPart2 = false;
        RayRec r0; uint32_t owner = 0;
        if (j < count) {
            if (TECH == T_LIGHT) live = light_step<false>(sc, cam, fr, st, io, j, r0);
            else if (TECH == T_NEE) { const int k = nee_consume(sc, cam, fr, st, io, j, owner); live = k == 1; toPart2 = k == 2; }
            else if (TECH == T_GI1) live = gi1_step(sc, cam, fr, st, io, j, r0, toPart2);
            else if (TECH == T_GI2) live = gi2_step(sc, cam, fr, st, io, j, r0);
            else live = path_step<(TECH <= T_BRDF ? TECH : T_BRUTE)>(sc, cam, fr, st, io, j, r0);
        }
        const uint32_t slot = block_append(live, io.countOut);
        if (live) {
            if (TECH == T_NEE) io.part2List[slot] = owner;             // NEE: the rays are built by k_nee_emit from this list
            else store_ray(io.raysOut, slot * io.raysPer, r0);
            if (TECH == T_GI2) io.ownersOut[slot] = io.ownersIn[j];
        }
        if (TECH == T_GI1) {
            const uint32_t slot2 = block_append(toPart2, io.part2Count);
            if (toPart2) io.part2List[slot2] = owner_of(io, j);
        }
        if (TECH == T_NEE) {                                           // BRDF rays that hit an emitter: MIS list (PathIO::misList)
            const uint32_t slot2 = block_append(toPart2, io.misCount);
            if (toPart2) io.misList[slot2] = owner;
        }
    }
}

}  // namespace rt
