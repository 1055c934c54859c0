// Device-side refit of the acceleration structure for moved vertices (dynamic scenes; the reference rebuilds
// every BLAS/TLAS on the host, SceneManager.cpp:83-129).  Topology (triangles, meshes, materials, tree shape) is kept; what
// changes with the vertices is recomputed where it lives:
//   k_refresh_triangles   per triangle: position record (triPos) and shading record (triShade) from the new vertices
//   k_refresh_leaf_tris   per leaf triangle: (v0, e1, e2) from the new positions
//   k_refit_level         per node of one level (levels = 1 first: all children are leaves): exact child boxes from the leaf
//                         triangles / the children's own boxes, then the node's grid (origin, power-of-two steps) and the 8-bit
//                         child planes with the host builder's rule (lo planes down, hi planes up, 1/16 step of slack,
//                         verified in fp32) — bvh_build.cpp: Collapser::quantise
// The tree keeps its shape, so its SAH quality degrades with large deformations; a full fyprt_upload_scene rebuilds it.
#pragma once
#include "rt_device.h"

namespace rt {

struct DevVertex { float px, py, pz, nx, ny, nz, u, v; };

__global__ void k_refresh_triangles(const DevVertex* verts, const uint4* triIdx, float4* triPos, float4* triShade, uint32_t nT) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nT) return;
    const uint4 ix = triIdx[t];
    const DevVertex a = verts[ix.x], b = verts[ix.y], c = verts[ix.z];
    const float mat = __int_as_float((int)ix.w);
    float4* p = triPos + (size_t)t * 3; float4* q = triShade + (size_t)t * 4;
    p[0] = make_float4(a.px, a.py, a.pz, mat); p[1] = make_float4(b.px, b.py, b.pz, 0.0f); p[2] = make_float4(c.px, c.py, c.pz, 0.0f);
    q[0] = make_float4(a.nx, a.ny, a.nz, a.u); q[1] = make_float4(b.nx, b.ny, b.nz, a.v); q[2] = make_float4(c.nx, c.ny, c.nz, b.u);
    q[3] = make_float4(b.v, c.u, c.v, mat);
}

// World vertices of one mesh from its object-space vertices and its model matrix — the vertex loop of Scene::AddNewMeshToScene /
// SceneManager::PerformAllSceneUpdates (Scene.cpp:42-51, SceneManager.cpp:30-41): position = (M * (p, 1)).xyz / w, normal =
// normalize((M * (n, 0)).xyz) (the model matrix, not its inverse transpose), texture coordinates unchanged.  Operation order as in
// host/HostTypes.h Mesh::ToWorld and scene.py Scene._to_world, so a transform edit uploads 64 bytes instead of the mesh's vertices.
struct Mat4 { float m[16]; };      // column-major
__global__ void k_transform_vertices(const DevVertex* obj, DevVertex* world, uint32_t first, uint32_t count, Mat4 M) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const DevVertex v = obj[first + k];
    const float* m = M.m;
    const float px = (m[0] * v.px + m[4] * v.py) + (m[8] * v.pz + m[12]);
    const float py = (m[1] * v.px + m[5] * v.py) + (m[9] * v.pz + m[13]);
    const float pz = (m[2] * v.px + m[6] * v.py) + (m[10] * v.pz + m[14]);
    const float pw = (m[3] * v.px + m[7] * v.py) + (m[11] * v.pz + m[15]);
    const float nx = (m[0] * v.nx + m[4] * v.ny) + (m[8] * v.nz);
    const float ny = (m[1] * v.nx + m[5] * v.ny) + (m[9] * v.nz);
    const float nz = (m[2] * v.nx + m[6] * v.ny) + (m[10] * v.nz);
    const float inv = 1.0f / __builtin_sqrtf((nx * nx + ny * ny) + nz * nz);
    DevVertex o; o.px = px / pw; o.py = py / pw; o.pz = pz / pw; o.nx = nx * inv; o.ny = ny * inv; o.nz = nz * inv; o.u = v.u; o.v = v.v;
    world[first + k] = o;
}

__global__ void k_refresh_leaf_tris(const float4* triPos, float4* leafTris, uint32_t nLeaf) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nLeaf) return;
    float4* r = leafTris + (size_t)j * 3;
    const float4 c = r[2];
    const uint32_t tri = (uint32_t)__float_as_int(c.y);
    const float4* p = triPos + (size_t)tri * 3;
    const float4 a = p[0], b = p[1], d = p[2];
    r[0] = make_float4(a.x, a.y, a.z, b.x - a.x);
    r[1] = make_float4(b.y - a.y, b.z - a.z, d.x - a.x, d.y - a.y);
    r[2] = make_float4(d.z - a.z, c.y, c.z, c.w);
}

// Sum of squared RGB differences of two ABGR8 images over rows [rowBegin, rowEnd) — MisUtils::ComputeMSE (MisUtils.cpp:118-147) as a
// device reduction: exact integer arithmetic (3 x 255^2 per pixel into a 64-bit sum), so the MSE equals the host routine's bit for bit.
// `flipRef`: the reference is read vertically flipped, as ComputeMSE reads its BMP-loaded original.
__global__ __launch_bounds__(256) void k_image_sqdiff(const uint32_t* image, const uint32_t* ref, uint32_t W, uint32_t H, uint32_t rowBegin, uint32_t rowEnd, int flipRef, unsigned long long* sum) {
    __shared__ unsigned long long s_part[4];
    unsigned long long acc = 0;
    const size_t n = (size_t)(rowEnd - rowBegin) * W;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const uint32_t y = rowBegin + (uint32_t)(k / W), x = (uint32_t)(k % W);
        const uint32_t a = image[(size_t)y * W + x], b = ref[(size_t)(flipRef ? H - 1u - y : y) * W + x];
#pragma unroll
        for (int sft = 0; sft < 24; sft += 8) { const int d = (int)((a >> sft) & 0xFFu) - (int)((b >> sft) & 0xFFu); acc += (unsigned long long)(d * d); }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0u) atomicAdd(sum, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

struct RBox { float lo[3], hi[3]; };
RT_DEV void rbox_grow(RBox& b, float x, float y, float z) {
    b.lo[0] = __builtin_fminf(b.lo[0], x); b.lo[1] = __builtin_fminf(b.lo[1], y); b.lo[2] = __builtin_fminf(b.lo[2], z);
    b.hi[0] = __builtin_fmaxf(b.hi[0], x); b.hi[1] = __builtin_fmaxf(b.hi[1], y); b.hi[2] = __builtin_fmaxf(b.hi[2], z);
}

// nodeBox: 2 float4 per node (lo.xyz, hi.xyz), written for every node processed
__global__ __launch_bounds__(128) void k_refit_level(float4* nodes, const uint32_t* levelNodes, uint32_t count, const float4* leafTris, const float4* triPos, float4* nodeBox) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const uint32_t node = levelNodes[k];
    float4* n = nodes + (size_t)node * 4;
    const float4 q0 = n[0], q1 = n[1];
    const uint32_t ex = (uint32_t)__float_as_int(q0.w), cnt = (ex >> 24) & 7u;
    const int32_t child[4] = {__float_as_int(q1.x), __float_as_int(q1.y), __float_as_int(q1.z), __float_as_int(q1.w)};
    RBox cb[4]; RBox nb;
    const float big = 3.402823466e+38f;
#pragma unroll
    for (int a = 0; a < 3; ++a) { nb.lo[a] = big; nb.hi[a] = -big; }
    // (every loop over the four child slots and the three axes is fully unrolled and predicated on `i < cnt`: the per-child boxes are
    // then indexed by constants and live in registers — indexed dynamically they were 112 bytes of scratch per lane)
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        RBox b;
#pragma unroll
        for (int a = 0; a < 3; ++a) { b.lo[a] = big; b.hi[a] = -big; }
        if (i >= cnt) { cb[i] = b; continue; }
        if (child[i] >= 0) {
            const size_t ci = (size_t)(child[i] >> 6);                // device form: byte offset of the child node (rt_host.h)
            const float4 l = nodeBox[ci * 2], h = nodeBox[ci * 2 + 1];
            b.lo[0] = l.x; b.lo[1] = l.y; b.lo[2] = l.z; b.hi[0] = h.x; b.hi[1] = h.y; b.hi[2] = h.z;
        } else {
            const uint32_t code = (uint32_t)~child[i], first = code >> 2, m = (code & 3u) + 1u;
            for (uint32_t j = 0; j < m; ++j) {
                const uint32_t tri = (uint32_t)__float_as_int(leafTris[(size_t)(first + j) * 3 + 2].y);
                const float4* p = triPos + (size_t)tri * 3;
                const float4 a = p[0], bb = p[1], c = p[2];
                rbox_grow(b, a.x, a.y, a.z); rbox_grow(b, bb.x, bb.y, bb.z); rbox_grow(b, c.x, c.y, c.z);
            }
        }
        cb[i] = b;
#pragma unroll
        for (int a = 0; a < 3; ++a) { nb.lo[a] = __builtin_fminf(nb.lo[a], b.lo[a]); nb.hi[a] = __builtin_fmaxf(nb.hi[a], b.hi[a]); }
    }
    uint32_t exps = ex & 0xFF000000u, qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = nb.lo[a];
        const double ext = (double)nb.hi[a] - (double)lo;
        int e = 1;
        if (ext > 0.0) { int ee; const double m = frexp(ext / 254.0, &ee); if (m == 0.5) --ee; e = ee + 127; e = e < 1 ? 1 : (e > 254 ? 254 : e); }
        uint32_t wl = 0, wh = 0;
        for (;; ++e) {
            const double s = ldexp(1.0, e - 127); const float sf = (float)s;
            bool ok = ext <= 254.0 * s;
            wl = 0; wh = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                int ql = 255, qh = 0;                                  // unused slots: the inverted box no ray can hit
                if (i < cnt && ok) {
                    ql = (int)floor(((double)cb[i].lo[a] - (double)lo) / s - 0.0625); qh = (int)ceil(((double)cb[i].hi[a] - (double)lo) / s + 0.0625);
                    ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                    while (ql > 0 && !(__builtin_fmaf((float)ql, sf, lo) <= cb[i].lo[a])) --ql;
                    while (qh < 255 && !(__builtin_fmaf((float)qh, sf, lo) >= cb[i].hi[a])) ++qh;
                    ok = __builtin_fmaf((float)ql, sf, lo) <= cb[i].lo[a] && __builtin_fmaf((float)qh, sf, lo) >= cb[i].hi[a];
                }
                wl |= (uint32_t)ql << (8 * i); wh |= (uint32_t)qh << (8 * i);
            }
            if (ok || e >= 254) break;
        }
        exps |= (uint32_t)e << (8 * a); qlo[a] = wl; qhi[a] = wh;
    }
    n[0] = make_float4(nb.lo[0], nb.lo[1], nb.lo[2], __int_as_float((int)exps));
    n[2] = make_float4(__int_as_float((int)qlo[0]), __int_as_float((int)qlo[1]), __int_as_float((int)qlo[2]), __int_as_float((int)qhi[0]));
    n[3] = make_float4(__int_as_float((int)qhi[1]), __int_as_float((int)qhi[2]), 0.0f, 0.0f);
    nodeBox[(size_t)node * 2] = make_float4(nb.lo[0], nb.lo[1], nb.lo[2], 0.0f);
    nodeBox[(size_t)node * 2 + 1] = make_float4(nb.hi[0], nb.hi[1], nb.hi[2], 0.0f);
}

// The reference keeps ONE pair of octahedral-normal buffers for both ReSTIRs (R.cu: normalBuffers prev / cur, swapped after every ReSTIR
// frame), so the "previous normal" a ReSTIR DI frame tests its history against is the last ReSTIR frame's — also when that was a GI
// frame — and vice versa.  Here the DI history carries its normal inside the 32-byte record and GI has its own normal buffers; when
// the technique changes between two ReSTIR frames this kernel carries the newest normals over, once (toRecords: GI buffer -> DI
// history records; else DI history records -> GI buffer).
__global__ void k_sync_history_normals(DIRec* records, f2* normals, uint32_t n, int toRecords) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4* q = reinterpret_cast<float4*>(records + i);
    if (toRecords) { const f2 nn = normals[i]; float4 a = q[0]; a.y = nn.x; a.z = nn.y; q[0] = a; }
    else { const float4 a = q[0]; f2 nn; nn.x = a.y; nn.y = a.z; normals[i] = nn; }
}

// fyprt_selftest_math: the three lean functions of rt_math.h against the compiler's correctly rounded sequences on every one of the
// 2^32 binary32 arguments.  counts[which] = number of arguments with different result bits (NaN == NaN), first[which] = the smallest such.
__global__ void k_math_selftest(int which, unsigned long long* counts, uint32_t* first) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t bad = 0, firstBad = 0xFFFFFFFFu;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        float a, b;
        if (which == 0) { a = sqrt_exact(x); b = __builtin_sqrtf(x); }
        else if (which == 1) { a = rcp_exact(x); b = 1.0f / x; }
        else { a = rsqrt_exact(x); b = 1.0f / __builtin_sqrtf(x); }
        if (!(__float_as_uint(a) == __float_as_uint(b) || (a != a && b != b))) { ++bad; firstBad = (uint32_t)i < firstBad ? (uint32_t)i : firstBad; }
    }
    if (bad) { atomicAdd(counts + which, (unsigned long long)bad); atomicMin(first + which, firstBad); }
}

}  // namespace rt
