// Shading functions of the reference's nine sampling techniques + the ReSTIR DI kernels (gfx950).
// Each function cites the reference code it computes; the arithmetic order is the reference's (Renderer.cu:565-2387), the
// execution structure is ours: primary rays one wave per 8x8 pixel tile (coherent), 4 waves per workgroup, traversal stack in
// LDS, scene read through the quad streams of rt_device.h, accumulation / tonemap / pack fused.  Techniques 0-6 and ReSTIR GI
// run as wavefront stages (rt_paths.h); ReSTIR DI as Part 1 + setup + persistent trace (rt_wavefront.h).
#pragma once
#include "rt_device.h"

namespace rt {

enum Tech { T_BRUTE = 0, T_UNIFORM = 1, T_COSINE = 2, T_GGX = 3, T_BRDF = 4, T_LIGHT = 5, T_NEE = 6, T_DI = 7, T_GI = 8 };

// pixel owned by this thread: workgroup = 16x16 pixels, wave = 8x8 tile
RT_DEV bool pixel_of_thread(const DevFrame& fr, uint32_t rowBegin, uint32_t rowEnd, uint32_t& x, uint32_t& y) {
    // Tile order (speed only; any placement is correct).  Workgroups are dealt round-robin over the 8 XCDs
    // (blockIdx % 8 share an L2).  0: linear.  1: each XCD gets one contiguous eighth of the tiles.
    // 2: each XCD gets every 8th ROW of tiles (L2 locality along a row, load balance down the image).
    // The grid is padded so every XCD owns the same number of workgroups.
    const uint32_t tilesX = (fr.W + 15u) >> 4;
    const uint32_t tilesY = (rowEnd - rowBegin + 15u) >> 4, nTiles = tilesX * tilesY;
    uint32_t bx, by;
    if (fr.tileOrder == 1u) {
        const uint32_t perXcd = gridDim.x >> 3;
        const uint32_t tile = (blockIdx.x & 7u) * perXcd + (blockIdx.x >> 3);
        if (tile >= nTiles) return false;
        bx = tile % tilesX; by = tile / tilesX;
    } else if (fr.tileOrder == 2u) {
        const uint32_t idx = blockIdx.x >> 3;
        bx = idx % tilesX; by = (idx / tilesX) * 8u + (blockIdx.x & 7u);
        if (by >= tilesY) return false;
    } else {
        if (blockIdx.x >= nTiles) return false;
        bx = blockIdx.x % tilesX; by = blockIdx.x / tilesX;
    }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    x = (bx << 4) + ((wave & 1u) << 3) + (lane & 7u);
    y = rowBegin + (by << 4) + ((wave >> 1) << 3) + (lane >> 3);
    return x < fr.W && y < rowEnd;
}

// Part 1 of ReSTIR: rows [p1Begin, p1End) plus, for a top band of a multi-GPU split, the single row `extraRow` (mapped to the
// first row of one more row of tiles; 0xFFFFFFFF = none)
RT_DEV bool p1_pixel_of_thread(const DevFrame& fr, uint32_t p1Begin, uint32_t p1End, uint32_t extraRow, uint32_t& x, uint32_t& y) {
    if (extraRow == 0xFFFFFFFFu) return pixel_of_thread(fr, p1Begin, p1End, x, y);
    if (!pixel_of_thread(fr, p1Begin, p1End + 16u, x, y)) return false;
    if (y < p1End) return true;
    if (y != ((p1End - p1Begin + 15u) & ~15u) + p1Begin) return false;      // only the first row of the extra tile row
    y = extraRow;
    return true;
}

// Common epilogue of all 11 reference kernels (e.g. Renderer.cu:2448-2465)
RT_DEV void epilogue(const DevFrame& fr, uint32_t i, f4 c) {
    if (!(finitef(c.x) && finitef(c.y) && finitef(c.z) && finitef(c.w))) c = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 a4 = fr.accum[i];
    f4 a = mk4(a4.x + c.x, a4.y + c.y, a4.z + c.z, a4.w + c.w);
    fr.accum[i] = make_float4(a.x, a.y, a.z, a.w);
    const float fi = (float)fr.frameIndex;
    a = mk4(a.x / fi, a.y / fi, a.z / fi, a.w / fi);
    a = mk4(a.x / (a.x + 1.0f), a.y / (a.y + 1.0f), a.z / (a.z + 1.0f), a.w / (a.w + 0.0f));
    a = mk4(gclamp(a.x, 0.0f, 1.0f), gclamp(a.y, 0.0f, 1.0f), gclamp(a.z, 0.0f, 1.0f), gclamp(a.w, 0.0f, 1.0f));
    fr.image[i] = pack_abgr(a);
}
RT_DEV f4 rgb1(f3 c) { return mk4(c.x, c.y, c.z, 1.0f); }

// ============================================================ techniques 0-4 (Renderer.cu:565-1284)
template <int TECH>
RT_DEV f3 sample_dir(f3 n, f3 V, const Mat& m, f3 albedo, float ggxRoughness, uint32_t& seed, float& pdf) {
    if (TECH == T_BRUTE || TECH == T_UNIFORM) { pdf = pdf_uniform(); return sample_uniform(n, seed); }
    if (TECH == T_COSINE) { pdf = -1.0f; return sample_cosine(n, seed); }
    if (TECH == T_GGX) return sample_ggx(n, V, ggxRoughness, seed, pdf);
    return sample_brdf(n, V, albedo, m.metallic, m.roughness, seed, pdf);
}

// ============================================================ light tree (LightTree.cuh:91-117, LightTree.cu, ConeBounds.cuh:47-87)
// LightTreeNode importance (LightTree.cuh:91-117) with FindConeThatEnvelopsAABBFromPoint (ConeBounds.cuh:47-87) inlined.
//  * The reference takes max_k acos(clamp(dot_k)) over the 8 box corners.  acos_f is monotone non-increasing over every float in
//    [-1, 1] (exhaustive check: tests/test_oracle_math.py), so that maximum is acos_f(min_k clamp(dot_k)) bit for bit: one binary64
//    acos per cluster instead of eight.  NaN corners are skipped by fmaxf there and by fminf here; the identity of the maximum (0)
//    is acos_f(1).
//  * The cone axis normalize(centroid - p) and the direction normalize(p - centroid) of the importance are each other's exact
//    negation (x - y == -(y - x), squares and the reciprocal square root are the same numbers), and d2 is the squared length the
//    normalisation computes anyway: one normalisation instead of two, every value bitwise what the reference computes.
// FLAT: the cluster's box has no extent along this axis (0 / 1 / 2; -1: none known) for EVERY lane of the wave — lights on a ceiling or a wall: the
// four corners that differ from the other four in that coordinate only are the same four points, their terms the same values, and a minimum does
// not care how often it sees a value: they are skipped (r03; the same bits, two thirds of the work — tests: every NEE / light-sampling parity test).
template <int FLAT>
RT_DEV float cluster_importance_t(f3 spPos, const DevLTNode& c) {
#define RT_CORNER_SKIPPED(k) (FLAT >= 0 && ((k) & (4 >> FLAT)) != 0)
    // nine reciprocal lengths — the centroid's and the eight corners' — are the bulk of the work.  They go through rt_math.h's lean
    // correctly rounded 1/sqrt (13 instructions instead of 27), guarded ONCE for the nine arguments with a wave-uniform branch, so
    // the nine chains still interleave (a guard per call serialises them: measured, −5 % instead of −20 % on the pick kernel).
    const f3 toC = mk3(c.centroid[0], c.centroid[1], c.centroid[2]) - spPos;
    const float dd = dot(toC, toC);
    f3 v[8]; float d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (RT_CORNER_SKIPPED(k)) continue;
        v[k] = mk3((k & 4) ? c.hi[0] : c.lo[0], (k & 2) ? c.hi[1] : c.lo[1], (k & 1) ? c.hi[2] : c.lo[2]) - spPos;
        d[k] = dot(v[k], v[k]);
    }
    uint32_t lo = __float_as_uint(dd), hi = lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (RT_CORNER_SKIPPED(k)) continue; const uint32_t b = __float_as_uint(d[k]); lo = b < lo ? b : lo; hi = b > hi ? b : hi; }
    float inv, invk[8];
    if (__ballot(!(lean_range(__uint_as_float(lo)) && lean_range(__uint_as_float(hi)))) == 0ull) {
        inv = lean_rcp(lean_sqrt(dd));
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (RT_CORNER_SKIPPED(k)) continue; invk[k] = lean_rcp(lean_sqrt(d[k])); }
    } else {
        inv = 1.0f / __builtin_sqrtf(dd);
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (RT_CORNER_SKIPPED(k)) continue; invk[k] = 1.0f / __builtin_sqrtf(d[k]); }
    }
    const f3 axis = toC * inv;                                  // == normalize(centroid - p)
    // min over k of clamp(dot_k, -1, 1), starting from 1 — evaluated as clamp(min(1, dot_0, ..., dot_7), -1, 1): clamping is monotonic
    // and passes every in-range value through unchanged, the running minimum never exceeds 1, and fminf skips a NaN dot exactly as it
    // skips the NaN its clamp would have been; so the result is the same bits with one clamp instead of eight.
    float minDot = 1.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (RT_CORNER_SKIPPED(k)) continue; minDot = __builtin_fminf(minDot, dot(axis, v[k] * invk[k])); }      // v * inv == normalize(corner - p)
    minDot = gclamp(minDot, -1.0f, 1.0f);
    const float theta_u = acos_f(minDot);
    const float d2 = __builtin_fmaxf(dd, 1e-12f);               // == dot(p - centroid, p - centroid)
    const f3 dir = -axis;                                       // == normalize(p - centroid)
    const float dotVal = gclamp(dot(mk3(c.axis[0], c.axis[1], c.axis[2]), dir), -1.0f, 1.0f);
    const float theta = acos_f(dotVal);
    const float angleTerm = gclamp((theta - c.theta_o) - theta_u, 0.0f, c.theta_e);
    return (c.energy * cos_f(angleTerm)) / d2;
}
#undef RT_CORNER_SKIPPED
RT_DEV float cluster_importance(f3 spPos, const DevLTNode& c) {
    // (wave-uniform: one lane whose box is not flat sends the whole wave down the general path; "flat" = the same BITS, so the skipped corners are
    // exact duplicates down to the sign of a zero)
    if (__ballot(__float_as_uint(c.lo[1]) != __float_as_uint(c.hi[1])) == 0ull) return cluster_importance_t<1>(spPos, c);
    if (__ballot(__float_as_uint(c.lo[0]) != __float_as_uint(c.hi[0])) == 0ull) return cluster_importance_t<0>(spPos, c);
    if (__ballot(__float_as_uint(c.lo[2]) != __float_as_uint(c.hi[2])) == 0ull) return cluster_importance_t<2>(spPos, c);
    return cluster_importance_t<-1>(spPos, c);
}
RT_DEV uint32_t lt_descend(const DevLTNode* nodes, uint32_t idx, f3 spPos, float& r, float& pmf) {
    const uint32_t l = nodes[idx].left, rt_ = nodes[idx].rightOrEmitter;
    float Il = cluster_importance(spPos, nodes[l]);
    const float Ir = cluster_importance(spPos, nodes[rt_]);
    float sum = Il + Ir;
    if (!(sum > 0.0f) || (Il + Ir) <= 0.0f) { sum = 1.0f; Il = 0.5f; }
    const float pl = gclamp(Il / sum, 1e-6f, 1.0f - 1e-6f);
    if (r < pl) { pmf *= pl; r = r / pl; return l; }
    const float pr = 1.0f - pl; pmf *= pr; r = (r - pl) / pr; return rt_;
}
struct PickedLight { uint32_t tri; float pmf; };
RT_DEV PickedLight pick_light(const DevScene& sc, f3 spPos, uint32_t& seed) {      // PickLight_TLAS + PickLight_BLAS
    PickedLight out; out.tri = ~0u; out.pmf = 0.0f;
    float r = rnd(seed);
    if (sc.ltTlasCount == 0 || sc.ltTlasRoot == ~0u) return out;
    uint32_t idx = sc.ltTlasRoot; float pmfT = 1.0f;
    r = gclamp(r, 0.0f, 0.9999999f);
    while (!sc.ltTlas[idx].isLeaf) idx = lt_descend(sc.ltTlas, idx, spPos, r, pmfT);
    const uint32_t mesh = sc.ltTlas[idx].rightOrEmitter;
    if (sc.ltCount[mesh] == 0 || sc.ltRoot[mesh] == ~0u) return out;
    const DevLTNode* b = sc.ltBlas + sc.ltFirst[mesh];
    uint32_t bi = sc.ltRoot[mesh]; float pmfB = 1.0f;
    r = gclamp(r, 0.0f, 0.9999999f);
    while (!b[bi].isLeaf) bi = lt_descend(b, bi, spPos, r, pmfB);
    out.tri = b[bi].rightOrEmitter; out.pmf = pmfT * pmfB;
    return out;
}
RT_DEV float direct_emitter_pmf(const DevScene& sc, f3 spPos, uint32_t emitterTri) {   // ComputeDirectEmitterPMF, LightTree.cu:156-276
    if (sc.ltTlasCount == 0 || sc.ltTlasRoot == ~0u) return 0.0f;
    const uint32_t tlasLeaf = emitterTri < sc.triCount ? sc.ltLeafOfTri[emitterTri] : ~0u;   // the reference's linear search (:170-199), tabled at upload
    if (tlasLeaf == ~0u) return 0.0f;
    float pmf = 1.0f; uint32_t idx = sc.ltTlasRoot;
    const DevLTNode* nodes = sc.ltTlas; uint32_t target = tlasLeaf;
    for (int level = 0; level < 2; ++level) {
        while (!nodes[idx].isLeaf) {                              // index-range heuristic (:227, :263), bug-for-bug
            const uint32_t l = nodes[idx].left, r = nodes[idx].rightOrEmitter;
            float Il = cluster_importance(spPos, nodes[l]);
            const float Ir = cluster_importance(spPos, nodes[r]);
            float sum = Il + Ir;
            if (!(sum > 0.0f)) { Il = 0.5f; sum = 1.0f; }
            const float pl = Il / sum;
            if (l <= target && target <= l + (nodes[l].numEmitters - 1u)) { pmf *= pl; idx = l; }
            else { pmf *= (1.0f - pl); idx = r; }
        }
        if (level == 0) {
            const uint32_t mesh = nodes[idx].rightOrEmitter;
            if (sc.ltCount[mesh] == 0) return 0.0f;
            nodes = sc.ltBlas + sc.ltFirst[mesh]; idx = sc.ltRoot[mesh]; target = emitterTri;
        }
    }
    return pmf;
}

// ============================================================ ReSTIR DI (Renderer.cu:1628-2041)
RT_DEV bool di_update(DIRes& r, uint32_t cand, float w, uint32_t count, float pdf, uint32_t& seed) {   // ReSTIR_DI_Reservoir.cu:3-36
    r.wSum += w; r.M += count;
    if (rnd(seed) < w / r.wSum) { r.index = cand; r.pdf = pdf; return true; }
    return false;
}
RT_DEV DIRes di_empty() { DIRes r; r.index = 0; r.W = 0.0f; r.pdf = 0.0f; r.wSum = 0.0f; r.M = 0; return r; }
// unshadowed target at the light centroid (Renderer.cu:1680-1730, :1799-1849)
// Per-emissive-slot light record, built once per scene upload ON THE DEVICE with the very expressions the reference
// evaluates per candidate (centroid Triangle.cuh:14-18, normal :36-43, 1/area :45-51, GetEmission Material.cu:5-8), so
// using it is bit-identical to recomputing — it removes the emissive[] -> triangle -> 3 vertices -> material gather
// chain (R.cu:1681-1688, :1722) and ~40 ALU ops from each of the 4+1 target evaluations per pixel.
//   q0 = centroid.xyz, 1/area | q1 = normal.xyz, triangle index | q2 = emission.xyz, 0
__global__ void k_build_light_records(DevScene sc, float4* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= sc.emissiveCount) return;
    const uint32_t tri = sc.emissive[k];
    const TriGeom g = load_tri(sc, tri);
    const f3 c = tri_centroid(g), n = tri_normal(g), em = emission(load_mat(sc, g.mat));
    out[3 * k + 0] = make_float4(c.x, c.y, c.z, 1.0f / tri_area(g));
    out[3 * k + 1] = make_float4(n.x, n.y, n.z, __int_as_float((int)tri));
    out[3 * k + 2] = make_float4(em.x, em.y, em.z, 0.0f);
}
// unshadowed target at the light centroid (Renderer.cu:1680-1730, :1799-1849)
RT_DEV float di_target(const DevScene& sc, uint32_t emissiveSlot, const Payload& pp, f3 pd, const Mat& hm, f3 albedo) {
    const float4* L = sc.lightRecs + (size_t)emissiveSlot * 3;
    const float4 l0 = L[0], l1 = L[1], l2 = L[2];
    const f3 ep = mk3(l0.x, l0.y, l0.z);
    f3 dir = ep - pos3(pp);
    const float dist = length(pos3(pp) - ep);
    dir = dir / dist;
    const f3 brdf = eval_brdf(nrm3(pp), -pd, dir, albedo, hm.metallic, hm.roughness);
    const float cx = gmax(dot(dir, nrm3(pp)), 0.0f);
    const float cy = gmax(dot(-dir, mk3(l1.x, l1.y, l1.z)), 0.0f);
    const float sa = l0.w * (dist * dist);
    const f3 Lr = (((brdf * cx) * cy) / sa) * mk3(l2.x, l2.y, l2.z);
    return length(Lr);
}
RT_DEV int f2i_sat(float f) { if (!(f == f)) return 0; if (f >= 2147483520.0f) return 2147483647; if (f <= -2147483648.0f) return (-2147483647 - 1); return (int)f; }
// `row` = the reprojected pixel's row: a row band only holds history of the rows it rendered last frame (DevFrame::histBegin/End)
RT_DEV uint32_t prev_pixel(const DevCamera& cam, f3 wp, uint32_t& row) {                    // Renderer.cu:1750-1763
    const f4 clip = mul(cam.prevProjView, mk4(wp.x, wp.y, wp.z, 1.0f));
    float nx = 0.0f, ny = 0.0f;
    if (clip.w != 0.0f) { nx = clip.x / clip.w; ny = clip.y / clip.w; }
    const float sx = (nx * 0.5f + 0.5f) * (float)cam.W, sy = (ny * 0.5f + 0.5f) * (float)cam.H;
    int px = f2i_sat(__builtin_floorf(sx)), py = f2i_sat(__builtin_floorf(sy));
    px = px < 0 ? 0 : (px > (int)cam.W - 1 ? (int)cam.W - 1 : px);
    py = py < 0 ? 0 : (py > (int)cam.H - 1 ? (int)cam.H - 1 : py);
    row = (uint32_t)py;
    return (uint32_t)py * cam.W + (uint32_t)px;
}
RT_DEV uint32_t neighbor_index(const DevCamera& cam, uint32_t W, uint32_t x, uint32_t y, uint32_t radius, uint32_t& seed) {   // :1915-1922
    float ox = 2.0f * rnd(seed) - 1.0f, oy = 2.0f * rnd(seed) - 1.0f;
    ox = (float)(uint32_t)(x + (uint32_t)(int)(ox * (float)radius));
    oy = (float)(uint32_t)(y + (uint32_t)(int)(oy * (float)radius));
    ox = __builtin_fmaxf(0.0f, __builtin_fminf((float)cam.W - 1.0f, ox));
    oy = __builtin_fmaxf(0.0f, __builtin_fminf((float)cam.H - 1.0f, oy));
    return (uint32_t)ox + (uint32_t)oy * W;
}

// Part 1 rows: [p1Begin, p1End) (band + halo); finished pixels (sky / emitter) go through the
// epilogue only inside the band proper so halo rows never touch accumulation.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_di_part1(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st, uint32_t p1Begin, uint32_t p1End, uint32_t extraRow) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch, + the top-node copy
    const float4* top4 = stage_top_nodes(sc.nodes, sc.topCount, sc.stackBudget, s_stack);
    uint32_t x, y;
    if (!p1_pixel_of_thread(fr, p1Begin, p1End, extraRow, x, y)) return;
    int32_t* stk = s_stack + threadIdx.x;
    const uint32_t i = x + y * fr.W;
    const bool inBand = (y >= fr.rowBegin && y < fr.rowEnd);
    uint32_t seed = i * (fr.frameIndex + 1u + st.randSeed);
    const f3 pd = ray_direction(cam, x, y);
    const Payload pp = trace_ray<COUNT>(sc, cam.position, pd, stk, top4);
    fr.payload[i] = pp;
    const f2 ncur = oct_encode(nrm3(pp));
    DIRes R = di_empty();
    bool finished = false; f3 finalColor = splat3(0.0f);
    Mat hm;
    if (pp.hitDistance < 0.0f) { finished = true; finalColor = st.sky; }
    else {
        hm = load_mat(sc, tri_material(sc, pp.objectIndex));
        if (length(emission(hm)) > 0.0f) { finished = true; finalColor = emission(hm); }
    }
    if (finished) {
        store_rec(fr.drec + i, pp.hitDistance, ncur, R);
        // history for the next frame: this frame's normal, the previous reservoir carried over unchanged (the
        // reference never writes di_prev_reservoirs for pixels finished in Part 1)
        store_rec(fr.dprevWrite + i, pp.hitDistance, ncur, rec_reservoir(load_rec(fr.dprevRead + i)));
        fr.depth[i] = pp.hitDistance;
        if (inBand && fr.p1Mode == 0u) epilogue(fr, i, rgb1(finalColor));
        return;
    }
    const f3 albedo = sample_albedo(sc, hm, pp.u, pp.v);
    const uint32_t nE = sc.emissiveCount;
    for (uint32_t k = 0; k < st.candidateCount; ++k) {
        const uint32_t e = (uint32_t)__builtin_roundf((float)(nE - 1u) * rnd(seed));
        const float pdf = di_target(sc, e, pp, pd, hm, albedo);
        di_update(R, e, pdf * (float)nE, 1u, pdf, seed);
    }
    R.W = R.pdf > 0.0f ? ((1.0f / R.pdf) * R.wSum) / (float)R.M : 0.0f;
    if (st.useTemporal) {
        uint32_t prow;
        const uint32_t prevIdx = prev_pixel(cam, pos3(pp), prow);
        const DIRec prec = load_rec(fr.dprevRead + prevIdx);
        f2 pn; pn.x = prec.nx; pn.y = prec.ny;
        const f3 prevN = oct_decode(pn);
        DIRes prev = rec_reservoir(prec);
        // history exists only for the rows this context rendered last frame (the whole frame unless fyprt_set_rows split it), and a
        // history light index must exist in the CURRENT emissive list (the scene may have been replaced: DESIGN.md §5 R7)
        const bool valid = (double)dot(prevN, nrm3(pp)) >= 0.99 && prow >= fr.histBegin && prow < fr.histEnd && prev.index < nE;
        if (valid && prev.M > 0u) {
            const uint32_t lim = st.historyLimit * R.M;
            prev.M = (lim < prev.M) ? lim : prev.M;
            DIRes Tm = di_empty(); uint32_t Z = 0;
            { const float pdf = R.pdf; di_update(Tm, R.index, (pdf * R.W) * (float)R.M, R.M, pdf, seed); Z += pdf > 0.0f ? R.M : 0u; }
            const float pdf = di_target(sc, prev.index, pp, pd, hm, albedo);
            di_update(Tm, prev.index, (pdf * prev.W) * (float)prev.M, prev.M, pdf, seed);
            Z += pdf > 0.0f ? prev.M : 0u;
            const float m = 1.0f / (float)Z;
            Tm.W = Tm.pdf > 0.0f ? (1.0f / Tm.pdf) * (m * Tm.wSum) : 0.0f;
            R = Tm;
        }
    }
    store_rec(fr.drec + i, pp.hitDistance, ncur, R);
    if (inBand && fr.p1Mode == 0u) fr.image[i] = 0u;     // sentinel: ConvertToRGBA(vec4(0)) (Renderer.cu:2746-2750)
}

template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_di_part2(DevScene sc, DevCamera cam, DevFrame fr, DevSettings st) {
    extern __shared__ int32_t s_stack[];                         // (stackBudget + 1) entries x kBlock threads, sized at launch
    uint32_t x, y;
    if (!pixel_of_thread(fr, fr.rowBegin, fr.rowEnd, x, y)) return;
    int32_t* stk = s_stack + threadIdx.x;
    const uint32_t i = x + y * fr.W;
    if (fr.image[i] != 0u) return;                                   // Renderer.cu:2787
    uint32_t seed = i * (fr.frameIndex + 213u + st.randSeed);
    const DIRec own = load_rec(fr.drec + i);
    DIRes R = rec_reservoir(own);
    const Payload pp = fr.payload[i];
    const Mat hm = load_mat(sc, tri_material(sc, pp.objectIndex));
    const f3 pd = ray_direction(cam, x, y);
    if (st.useSpatial) {
        uint32_t Z = 0; DIRes S = di_empty();
        { const float pdf = R.pdf; di_update(S, R.index, (pdf * R.W) * (float)R.M, R.M, pdf, seed); Z += pdf > 0.0f ? R.M : 0u; }
        for (uint32_t n = 0; n < st.numNeighbors; ++n) {
            const uint32_t ni = neighbor_index(cam, fr.W, x, y, st.radius, seed);
            const DIRec nb = load_rec(fr.drec + ni);
            const float nd = nb.hitDistance, pdp = pp.hitDistance;
            f2 nn; nn.x = nb.nx; nn.y = nb.ny;
            if ((nd > 1.1f * pdp || nd < 0.9f * pdp) || (double)dot(nrm3(pp), oct_decode(nn)) < 0.906) continue;
            const DIRes N = rec_reservoir(nb);
            const float pdf = N.pdf;
            di_update(S, N.index, (pdf * N.W) * (float)N.M, N.M, pdf, seed);
            Z += pdf > 0.0f ? N.M : 0u;
        }
        const float m = 1.0f / (float)Z;
        S.W = S.pdf > 0.0f ? (1.0f / S.pdf) * (m * S.wSum) : 0.0f;
        R = S;
    }
    const uint32_t ti = sc.emissive[R.index];
    const TriGeom g = load_tri(sc, ti);
    const f3 ep = tri_random_point(g, seed);
    f3 dir = ep - pos3(pp);
    const float dist = length(pos3(pp) - ep);
    dir = dir / dist;
    const f3 albedo = sample_albedo(sc, hm, pp.u, pp.v);
    const f3 brdf = eval_brdf(nrm3(pp), -pd, dir, albedo, hm.metallic, hm.roughness);
    const float cx = gmax(dot(dir, nrm3(pp)), 0.0f);
    const float cy = gmax(dot(-dir, tri_normal(g)), 0.0f);
    const float triAreaPDF = 1.0f / tri_area(g);
    const float sa = triAreaPDF * (dist * dist);
    const f3 T = ((brdf * cx) * cy) / sa;
    f3 Lvis = splat3(0.0f);                                          // the pixel's radiance if the light is visible / if the ray escapes
    { const Mat lm = load_mat(sc, g.mat); if (length(emission(lm)) > 0.0f) { Lvis = T * emission(lm); Lvis = Lvis * R.W; } }
    const f3 Lsky = T * st.sky;
    f3 radiance = splat3(0.0f);
    if (!(st.skipDeadRays && zero3(Lvis) && zero3(Lsky))) {          // (black in every outcome: no ray, DevSettings::skipDeadRays)
        const ShadowHit hit = trace_shadow<COUNT>(sc, pos3(pp) + nrm3(pp) * 1e-12f, dir, ti, stk);
        if ((uint32_t)hit.objectIndex == ti && hit.hitDistance >= 0.0f) radiance = Lvis;
        else if (hit.hitDistance < 0.0f) radiance = Lsky;
    }
    fr.depth[i] = pp.hitDistance;
    { f2 on; on.x = own.nx; on.y = own.ny; store_rec(fr.dprevWrite + i, pp.hitDistance, on, R); }
    epilogue(fr, i, rgb1(radiance));
}

// ============================================================ ReSTIR GI (Renderer.cu:2043-2387)
RT_DEV void gi_reset_sample(GISample& s) { s.vp[0] = s.vp[1] = s.vp[2] = 0.0f; s.vn[0] = s.vn[1] = 0.0f; s.sp[0] = s.sp[1] = s.sp[2] = 0.0f; s.sn[0] = s.sn[1] = 0.0f; s.Lo[0] = s.Lo[1] = s.Lo[2] = 0.0f; s.seed = 0; s.pdf = 0.0f; }
RT_DEV void gi_reset(GIRes& r) { gi_reset_sample(r.s); r.W = 0.0f; r.M = 0; r.wSum = 0.0f; r.pad[0] = r.pad[1] = 0.0f; }
RT_DEV bool gi_update(GIRes& r, const GISample& s, float w, uint32_t count, float pdf, uint32_t& seed) {   // ReSTIR_GI_Reservoir.cu:5-34
    r.wSum += w; r.M += count;
    if (rnd(seed) < w / r.wSum) { r.s = s; r.s.pdf = pdf; return true; }
    return false;
}
RT_DEV bool gi_merge(GIRes& r, const GIRes& o, float pdf, uint32_t& seed) {                                // :36-43; true: r's sample was replaced
    const uint32_t prev = r.M;
    const bool replaced = gi_update(r, o.s, (pdf * o.wSum) * (float)o.M, 1u, pdf, seed);
    r.M = prev + o.M;
    return replaced;
}
RT_DEV f3 lo3(const GISample& s) { return mk3(s.Lo[0], s.Lo[1], s.Lo[2]); }

}  // namespace rt
