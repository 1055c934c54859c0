// C-callable face of the host-side asset readers (TextureIO.h) for the Python mirror (fypraytracer_amd/texture.py):
// libfyprt_host.so, host code only — no GPU, no HIP.
#include "HostTypes.h"
extern "C" {
// Two calls: with pixels == nullptr it decodes, caches and reports the size; with a buffer of width*height words it copies.
int fyprt_host_load_png(const char* path, uint32_t* width, uint32_t* height, uint32_t* pixels, char* err, size_t errLen) {
    static thread_local fyprt_host::TextureImage img; static thread_local std::string cached;
    if (!pixels || cached != path) {
        cached.clear();
        if (!fyprt_host::LoadTexturePNG(path, img)) { if (err && errLen) { std::snprintf(err, errLen, "%s", img.error.c_str()); } return 1; }
        cached = path;
    }
    if (width) *width = img.width;
    if (height) *height = img.height;
    if (pixels) std::memcpy(pixels, img.pixels.data(), img.pixels.size() * 4);
    return 0;
}

// ---- the host producers of the kernel inputs (HostTypes.h) for the Python mirror (fypraytracer_amd/scene.py): ONE implementation of
// Mesh::UpdateWorldTransform, the vertex loop of Scene::AddNewMeshToScene / SceneManager, the emissive test and the Camera, used by the
// C++ facade directly and by Python through these entry points (no second hand-maintained copy of the arithmetic).
void fyprt_host_mesh_matrix(const float* pos3, const float* rot3, const float* scale3, float* out16) {
    fyprt_host::Mesh m; m.position = {pos3[0], pos3[1], pos3[2]}; m.rotation = {rot3[0], rot3[1], rot3[2]}; m.scale = {scale3[0], scale3[1], scale3[2]};
    fyprt_host::Mesh::UpdateWorldTransform(m);
    std::memcpy(out16, m.worldTransformMatrix.m, 64);
}
// vertices: n records of 8 floats (position, normal, uv: the reference's Vertex); uv passes through
void fyprt_host_to_world(const float* matrix16, const float* verticesIn, float* verticesOut, uint32_t n) {
    fyprt_host::Mesh m; std::memcpy(m.worldTransformMatrix.m, matrix16, 64);
    const fyprt_host::Vertex* in = reinterpret_cast<const fyprt_host::Vertex*>(verticesIn);
    fyprt_host::Vertex* out = reinterpret_cast<fyprt_host::Vertex*>(verticesOut);
    for (uint32_t i = 0; i < n; ++i) out[i] = m.ToWorld(in[i]);
}
int fyprt_host_is_emissive(const float* color3, float power) {                 // the test of Scene::InitSceneEmissiveTriangles (Scene.cpp:216)
    const float ex = color3[0] * power, ey = color3[1] * power, ez = color3[2] * power;
    return (ex * ex + ey * ey + ez * ez > 0.0f) ? 1 : 0;
}
void* fyprt_host_camera_create(float fov, float nearClip, float farClip) { return new fyprt_host::Camera(fov, nearClip, farClip); }
void fyprt_host_camera_destroy(void* c) { delete static_cast<fyprt_host::Camera*>(c); }
void fyprt_host_camera_on_resize(void* c, uint32_t w, uint32_t h) { static_cast<fyprt_host::Camera*>(c)->OnResize(w, h); }
void fyprt_host_camera_set_position(void* c, const float* p) { static_cast<fyprt_host::Camera*>(c)->SetPosition({p[0], p[1], p[2]}); }
void fyprt_host_camera_set_direction(void* c, const float* d) { static_cast<fyprt_host::Camera*>(c)->SetDirection({d[0], d[1], d[2]}); }
void fyprt_host_camera_assign_forward(void* c, const float* d) { static_cast<fyprt_host::Camera*>(c)->GetForwardDirection() = {d[0], d[1], d[2]}; }
void fyprt_host_camera_assign_position(void* c, const float* p) { static_cast<fyprt_host::Camera*>(c)->GetPosition() = {p[0], p[1], p[2]}; }
int fyprt_host_camera_on_update(void* c, float ts, const char* keys, float dx, float dy) { return static_cast<fyprt_host::Camera*>(c)->OnUpdate(ts, keys, {dx, dy}) ? 1 : 0; }
void fyprt_host_camera_commit_frame(void* c) { static_cast<fyprt_host::Camera*>(c)->CommitFrame(); }
// out: projection, view, prevProjection, prevView, inverseProjection, inverseView (16 floats each), position (3), forward (3)
void fyprt_host_camera_state(void* cv, float* out102) {
    fyprt_host::Camera& c = *static_cast<fyprt_host::Camera*>(cv);
    const fyprt_host::mat4* ms[6] = {&c.GetProjection(), &c.GetView(), &c.GetPrevProjection(), &c.GetPrevView(), &c.GetInverseProjection(), &c.GetInverseView()};
    for (int k = 0; k < 6; ++k) std::memcpy(out102 + 16 * k, ms[k]->m, 64);
    const fyprt_host::vec3 p = c.GetPosition(), f = c.GetForwardDirection();
    out102[96] = p.x; out102[97] = p.y; out102[98] = p.z; out102[99] = f.x; out102[100] = f.y; out102[101] = f.z;
}
}
