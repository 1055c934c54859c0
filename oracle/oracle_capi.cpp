// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle_vec.h header).
// extern "C" surface of the CPU restatement, loaded with ctypes by tests/, smoke() and
// bench.py's cpu_baseline leg.  Struct layouts mirror include/fyprt.h so the same ctypes
// structures describe a scene to both the checker and the product.
#include <omp.h>
#include <memory>
#include "oracle_product_trace.h"

using namespace orc;

#include <cstring>
extern "C" {

struct orc_vertex { float position[3], normal[3], uv[2]; };
struct orc_material { uint32_t is_use_albedo_map; float albedo[3]; uint32_t albedo_map_index; float roughness, metallic; float emission_color[3]; float emission_power; };
struct orc_mesh { uint32_t first_triangle, triangle_count; int32_t material_index; };
struct orc_texture { const uint32_t* pixels; uint32_t width, height; };
struct orc_lt_node { float energy; uint32_t num_emitters, left, right_or_emitter, is_leaf; float cone_axis[3], theta_o, theta_e; float box_lo[3], box_hi[3], box_centroid[3]; uint32_t _pad; };
struct orc_scene_desc {
    const orc_vertex* vertices; uint32_t vertex_count;
    const void* triangles; uint32_t triangle_count, triangle_stride;
    const orc_material* materials; uint32_t material_count;
    const orc_mesh* meshes; uint32_t mesh_count;
    const orc_texture* textures; uint32_t texture_count;
    const uint32_t* emissive_triangles; uint32_t emissive_count;
    const void* light_trees;
};
struct orc_camera_desc { float projection[16], view[16], prev_projection[16], prev_view[16], inverse_projection[16], inverse_view[16]; float position[3]; uint32_t viewport_width, viewport_height; };

struct orc_handle {
    Scene scene;
    std::unique_ptr<ReferenceTracer> refTracer;
    std::unique_ptr<ProductTracer> prodTracer;
    std::unique_ptr<Renderer> rr, rp;   // renderer bound to each tracer; share one Frame via swap
    bool useProduct = false;
    Renderer& R() { return useProduct ? *rp : *rr; }
};

orc_handle* orc_create(const orc_scene_desc* d) {
    auto* h = new orc_handle();
    Scene& s = h->scene;
    s.worldVertices.resize(d->vertex_count);
    for (uint32_t i = 0; i < d->vertex_count; ++i) {
        const orc_vertex& v = d->vertices[i];
        s.worldVertices[i] = Vertex{v3(v.position[0], v.position[1], v.position[2]), v3(v.normal[0], v.normal[1], v.normal[2]), vec2{v.uv[0], v.uv[1]}};
    }
    s.triangles.resize(d->triangle_count);
    for (uint32_t i = 0; i < d->triangle_count; ++i)
        std::memcpy(&s.triangles[i], (const char*)d->triangles + (size_t)i * d->triangle_stride, sizeof(TriIdx));
    s.materials.resize(d->material_count);
    for (uint32_t i = 0; i < d->material_count; ++i) {
        const orc_material& m = d->materials[i];
        s.materials[i] = Material{m.is_use_albedo_map & 0xFFu, v3(m.albedo[0], m.albedo[1], m.albedo[2]), m.albedo_map_index, m.roughness, m.metallic,
                                  v3(m.emission_color[0], m.emission_color[1], m.emission_color[2]), m.emission_power};
    }
    s.meshes.resize(d->mesh_count);
    for (uint32_t i = 0; i < d->mesh_count; ++i) s.meshes[i] = MeshRange{d->meshes[i].first_triangle, d->meshes[i].triangle_count, d->meshes[i].material_index};
    s.texturePixels.resize(d->texture_count); s.textures.resize(d->texture_count);
    for (uint32_t i = 0; i < d->texture_count; ++i) {
        const orc_texture& t = d->textures[i];
        s.texturePixels[i].assign(t.pixels, t.pixels + (size_t)t.width * t.height);
        s.textures[i] = Texture{s.texturePixels[i].data(), t.width, t.height};
    }
    s.Build();
    h->refTracer.reset(new ReferenceTracer(s));
    h->prodTracer.reset(new ProductTracer(s));
    h->rr.reset(new Renderer(s, *h->refTracer));
    h->rp.reset(new Renderer(s, *h->prodTracer));
    return h;
}
void orc_destroy(orc_handle* h) { delete h; }

uint32_t orc_emissive_count(orc_handle* h) { return (uint32_t)h->scene.emissiveTriangles.size(); }
void orc_get_emissive(orc_handle* h, uint32_t* dst) { std::memcpy(dst, h->scene.emissiveTriangles.data(), h->scene.emissiveTriangles.size() * 4); }

void orc_resize(orc_handle* h, uint32_t w, uint32_t hh) { h->rr->fr.Resize(w, hh); h->rp->fr = Frame(); }
static void bind(orc_handle* h, bool product) {      // move the frame state to the renderer that is about to run
    if (h->useProduct == product) return;
    Renderer& from = h->R(); h->useProduct = product; Renderer& to = h->R();
    to.fr = std::move(from.fr); to.cam = from.cam; from.fr = Frame();
}
void orc_use_reference_tracer(orc_handle* h) { bind(h, false); }
void orc_set_product_bvh(orc_handle* h, const void* nodes, uint32_t nodeCount, const void* tris, uint32_t triCount, int32_t rootRef) {
    ProductTracer& p = *h->prodTracer;
    p.nodes.resize(nodeCount); if (nodeCount) std::memcpy(p.nodes.data(), nodes, (size_t)nodeCount * 64);
    p.tris.resize(triCount); if (triCount) std::memcpy(p.tris.data(), tris, (size_t)triCount * 48);
    p.rootRef = rootRef;
    bind(h, true);
}
// deepest traversal stack the product-order restatement has used so far, and a knob to force its resume-entry path
int orc_product_max_stack(orc_handle* h) { return h->prodTracer->maxTop; }
void orc_set_product_stack_budget(orc_handle* h, int budget) { h->prodTracer->stackBudget = budget; }
void orc_set_skip_dead_rays(orc_handle* h, int on) { h->rp->skipDeadShadowRays = on != 0; }     // product twin only
// event log of the product-order traversal (tools/wave_sim.py): start with on = 1 (render with ONE thread), read with orc_take_events
static std::vector<uint8_t> g_events;
void orc_record_events(orc_handle* h, int on) { g_events.clear(); h->prodTracer->events = on ? &g_events : nullptr; }
size_t orc_take_events(orc_handle* h, uint8_t* dst, size_t cap) { (void)h; if (dst) std::memcpy(dst, g_events.data(), g_events.size() < cap ? g_events.size() : cap); return g_events.size(); }
void orc_set_camera(orc_handle* h, const orc_camera_desc* c) {
    Camera cam;
    cam.projection = mat4_from(c->projection); cam.view = mat4_from(c->view);
    cam.prevProjection = mat4_from(c->prev_projection); cam.prevView = mat4_from(c->prev_view);
    cam.inverseProjection = mat4_from(c->inverse_projection); cam.inverseView = mat4_from(c->inverse_view);
    cam.position = v3(c->position[0], c->position[1], c->position[2]);
    cam.width = c->viewport_width; cam.height = c->viewport_height;
    h->rr->cam = cam; h->rp->cam = cam;
}
// The per-pixel state (accumulation, image, reservoirs, history, normals, frame index) of `src` moves into `dst`: what the reference
// does when the application replaces or edits the scene on a live renderer — SceneToGPU / FreeSceneGPU touch the scene only, every
// per-pixel buffer stays (Renderer.cu:286-419 allocate them on a resize alone).  Both handles must have the same frame size.
void orc_adopt_frame(orc_handle* dst, orc_handle* src) { dst->R().fr = std::move(src->R().fr); src->R().fr = Frame(); }
void orc_reset_frame_index(orc_handle* h) { h->R().fr.frameIndex = 1; }
uint32_t orc_frame_index(orc_handle* h) { return h->R().fr.frameIndex; }
void orc_set_threads(int n) { omp_set_num_threads(n); }
int orc_max_threads() { return omp_get_max_threads(); }

// counters_out: rays, boxTests, triTests, hits
void orc_render(orc_handle* h, const void* settings52, uint32_t row_begin, uint32_t row_end, uint32_t halo, uint64_t* counters_out) {
    Renderer& R = h->R();
    std::memcpy(&R.st, settings52, 52);
    if (row_end > R.fr.H) row_end = R.fr.H;
    Counters c = R.RenderFrame(row_begin, row_end, halo);
    if (counters_out) { counters_out[0] = c.rays; counters_out[1] = c.boxTests; counters_out[2] = c.triTests; counters_out[3] = c.hits; counters_out[4] = c.nodeVisits; }
}

// which follows enum fyprt_buffer; returns bytes copied (0 on error)
size_t orc_read_buffer(orc_handle* h, int which, void* dst, size_t bytes) {
    Frame& f = h->R().fr; const void* src = nullptr; size_t n = 0;
    switch (which) {
        case 0: src = f.accum.data(); n = f.accum.size() * 16; break;
        case 1: src = f.image.data(); n = f.image.size() * 4; break;
        case 2: src = f.payload.data(); n = f.payload.size() * 40; break;
        case 3: src = f.depth.data(); n = f.depth.size() * 4; break;
        case 4: src = f.normalPrev.data(); n = f.normalPrev.size() * 8; break;   // after the end-of-frame swap "prev" holds the frame just rendered
        case 5: src = f.di.data(); n = f.di.size() * 20; break;
        case 6: src = f.diPrev.data(); n = f.diPrev.size() * 20; break;
        case 7: src = f.gi.data(); n = f.gi.size() * 72; break;
        case 8: src = f.giPrev.data(); n = f.giPrev.size() * 72; break;
        default: return 0;
    }
    if (bytes < n) n = bytes;
    std::memcpy(dst, src, n);
    return n;
}

// ---- reference-format structures, for tests of the product's builders and as prebuilt input (fyprt_lighttrees)
static void flat(const LTNode& n, orc_lt_node& o) {
    std::memset(&o, 0, sizeof o);
    o.energy = n.energy; o.num_emitters = n.numEmitters; o.left = n.offset; o.right_or_emitter = n.emitterIndex; o.is_leaf = n.isLeaf ? 1u : 0u;
    o.cone_axis[0] = n.bounds_o.axis.x; o.cone_axis[1] = n.bounds_o.axis.y; o.cone_axis[2] = n.bounds_o.axis.z; o.theta_o = n.bounds_o.theta_o; o.theta_e = n.bounds_o.theta_e;
    o.box_lo[0] = n.bounds_w.lo.x; o.box_lo[1] = n.bounds_w.lo.y; o.box_lo[2] = n.bounds_w.lo.z;
    o.box_hi[0] = n.bounds_w.hi.x; o.box_hi[1] = n.bounds_w.hi.y; o.box_hi[2] = n.bounds_w.hi.z;
    o.box_centroid[0] = n.bounds_w.centroid.x; o.box_centroid[1] = n.bounds_w.centroid.y; o.box_centroid[2] = n.bounds_w.centroid.z;
}
uint32_t orc_lighttree_tlas_count(orc_handle* h) { return (uint32_t)h->scene.lightTlas.nodes.size(); }
uint32_t orc_lighttree_blas_total(orc_handle* h) { uint32_t n = 0; for (auto& t : h->scene.lightBlas) n += (uint32_t)t.nodes.size(); return n; }
void orc_export_lighttrees(orc_handle* h, orc_lt_node* tlas, uint32_t* tlas_root, orc_lt_node* blas, uint32_t* first, uint32_t* count, uint32_t* root) {
    const Scene& s = h->scene;
    for (size_t i = 0; i < s.lightTlas.nodes.size(); ++i) flat(s.lightTlas.nodes[i], tlas[i]);
    *tlas_root = s.lightTlas.rootIndex;
    uint32_t off = 0;
    for (size_t m = 0; m < s.lightBlas.size(); ++m) {
        first[m] = off; count[m] = (uint32_t)s.lightBlas[m].nodes.size(); root[m] = s.lightBlas[m].rootIndex;
        for (auto& n : s.lightBlas[m].nodes) flat(n, blas[off++]);
    }
}
// reference BVH statistics (node count incl. all BLAS + TLAS, max depth) for reporting
void orc_reference_bvh_stats(orc_handle* h, uint64_t* nodes, uint32_t* tlasNodes) {
    uint64_t n = h->scene.tlas.nodes.size(); for (auto& b : h->scene.blas) n += b.nodes.size();
    *nodes = n; *tlasNodes = (uint32_t)h->scene.tlas.nodes.size();
}

// ---- single-ray probe: payload (40 B) of one TraceRay with the active tracer
void orc_trace(orc_handle* h, const float* origin3, const float* dir3, void* payload40, uint64_t* counters_out) {
    Ray r{v3(origin3[0], origin3[1], origin3[2]), v3(dir3[0], dir3[1], dir3[2])};
    Counters c; Payload p = h->R().tracer.Trace(r, c);
    std::memcpy(payload40, &p, 40);
    if (counters_out) { counters_out[0] = c.rays; counters_out[1] = c.boxTests; counters_out[2] = c.triTests; counters_out[3] = c.hits; counters_out[4] = c.nodeVisits; }
}
void orc_ray_direction(orc_handle* h, uint32_t x, uint32_t y, float* out3) { vec3 d = h->R().RayDirection(x, y); out3[0] = d.x; out3[1] = d.y; out3[2] = d.z; }

// ---- known-answer hooks (SURVEY.md §8c closed-form pins)
uint32_t orc_pcg_hash(uint32_t x) { return pcg_hash(x); }
float orc_random_float(uint32_t* seed) { return randomFloat(*seed); }
float orc_sin(float x) { return t_sin(x); }
float orc_cos(float x) { return t_cos(x); }
float orc_acos(float x) { return t_acos(x); }
// Exhaustive check that t_acos is monotone non-increasing over every float in [-1, 1] (2^31 + 1 values): the product
// evaluates max_k acos(x_k) of ConeThetaToBox (ConeBounds.cuh:47-87) as acos(min_k x_k), which is the same number iff
// this returns 0.  Returns the number of adjacent pairs (a < b) with t_acos(a) < t_acos(b).
uint64_t orc_acos_monotone_violations() {
    // order-preserving map: key k in [0, 2^31] -> float; k < 2^30+... handled through the sign-magnitude bit pattern
    const uint32_t one = 0x3F800000u;                       // bits of 1.0f; floats in [-1, 0) are keys one-1 .. 0 reversed
    const int64_t n = (int64_t)one * 2 + 1;                 // -1 .. -0 (one+1 values) then +0 .. 1 (one+1 values), minus the shared step
    auto at = [&](int64_t k) -> float {                     // k = 0 -> -1.0f, k = one -> -0.0f, k = one+1 -> +0.0f, k = 2*one+1 -> 1.0f
        uint32_t bits = (k <= (int64_t)one) ? (0x80000000u | (uint32_t)((int64_t)one - k)) : (uint32_t)(k - (int64_t)one - 1);
        float f; memcpy(&f, &bits, 4); return f;
    };
    uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t c = 0; c < 4096; ++c) {
        const int64_t lo = (n + 1) * c / 4096, hi = (n + 1) * (c + 1) / 4096;   // pairs (k, k+1), k in [lo, hi)
        float prev = t_acos(at(lo));
        for (int64_t k = lo; k < hi && k < n; ++k) {
            const float cur = t_acos(at(k + 1));
            if (prev < cur || cur != cur) ++bad;
            prev = cur;
        }
    }
    return bad;
}
float orc_pow5(float x) { return t_pow5(x); }
float orc_uniform_pdf() { return UniformHemispherePDF(); }
void orc_encode_oct(const float* n3, float* e2) { vec2 e = EncodeOctahedral(v3(n3[0], n3[1], n3[2])); e2[0] = e.x; e2[1] = e.y; }
void orc_decode_oct(const float* e2, float* n3) { vec3 n = DecodeOctahedral(vec2{e2[0], e2[1]}); n3[0] = n.x; n3[1] = n.y; n3[2] = n.z; }
uint32_t orc_convert_rgba(const float* c4) { return ConvertToRGBA(vec4{c4[0], c4[1], c4[2], c4[3]}); }
void orc_brdf(const float* N, const float* V, const float* L, const float* albedo, float metallic, float roughness, float* out3) {
    vec3 r = CalculateBRDF(v3(N[0], N[1], N[2]), v3(V[0], V[1], V[2]), v3(L[0], L[1], L[2]), v3(albedo[0], albedo[1], albedo[2]), metallic, roughness);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
// sampler probe: kind 0 uniform, 1 cosine, 2 ggx, 3 brdf-mixture; returns direction + pdf, advances seed
void orc_sample(int kind, const float* N, const float* V, const float* albedo, float metallic, float roughness, uint32_t* seed, float* out4) {
    vec3 n = v3(N[0], N[1], N[2]), v = v3(V[0], V[1], V[2]), a = v3(albedo[0], albedo[1], albedo[2]); float pdf = 0.0f; vec3 d;
    if (kind == 0) { d = UniformSampleHemisphere(n, *seed); pdf = UniformHemispherePDF(); }
    else if (kind == 1) { d = CosineSampleHemisphere(n, *seed); pdf = CosineHemispherePDF(gmax(dot(d, n), 0.0f)); }
    else if (kind == 2) d = GGXSampleHemisphere(n, v, roughness, *seed, pdf);
    else d = BRDFSampleHemisphere(n, v, a, metallic, roughness, *seed, pdf);
    out4[0] = d.x; out4[1] = d.y; out4[2] = d.z; out4[3] = pdf;
}
// DI reservoir KAT: reset + one update
void orc_di_reset_update(uint32_t cand, float w, float pdf, uint32_t* seed, void* out20, int* accepted) {
    DIReservoir r; DI_Reset(r); *accepted = DI_Update(r, cand, w, 1, pdf, *seed) ? 1 : 0; std::memcpy(out20, &r, 20);
}
uint32_t orc_sample_bilinear(const uint32_t* pixels, uint32_t w, uint32_t hh, float u, float v) { return SampleBilinear(Texture{pixels, w, hh}, u, v); }

}  // extern "C"
