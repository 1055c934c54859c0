// glm-free stand-ins with the reference's member names (Vertex.h, Triangle.cuh, Material.cuh, Mesh.h,
// Texture.cuh, Scene.h, Camera.h) — just enough of the scene API to drive the facade headlessly.
// Byte layouts equal the reference's (Vertex 32, Triangle 52, Material 44).
#pragma once
#include "TextureIO.h"
#include <cmath>
#include <cstdint>
#include <vector>

namespace fyprt_host {

struct vec2 { float x = 0, y = 0; };
struct vec3 { float x = 0, y = 0, z = 0; };
struct mat4 { float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; };   // column-major

struct Vertex { vec3 position, normal; vec2 uv; };
struct AABB { vec3 lowerBound, upperBound, centroidPos; };
struct Triangle { uint32_t v0, v1, v2; int materialIndex = 0; AABB aabb; };
struct Material {
    bool isUseAlbedoMap = false; vec3 albedo{1.0f, 0.0f, 1.0f}; uint32_t albedoMapIndex = 0xFFFFFFFFu;
    float roughness = 1.0f, metallic = 0.0f; vec3 emissionColor; float emissionPower = 0.0f;
};
struct Texture { uint32_t* pixels = nullptr; uint32_t width = 0, height = 0; std::string fileName; };
struct Mesh {                                   // Mesh.h: ranges into the scene arrays + the transform the UI edits
    uint32_t vertexStart = 0, vertexCount = 0, indexStart = 0, indexCount = 0; int materialIndex = 0;
    vec3 position{0, 0, 0}, rotation{0, 0, 0}, scale{1, 1, 1}; mat4 worldTransformMatrix;
    // Mesh::UpdateWorldTransform: T * yawPitchRoll(radians(rot.y), radians(rot.x), radians(rot.z)) * S (glm::yawPitchRoll,
    // angles and trigonometry in double, entries rounded to float; products in float, column by column)
    static void UpdateWorldTransform(Mesh& m) {
        const double d2r = 3.14159265358979323846 / 180.0;
        const double yaw = m.rotation.y * d2r, pitch = m.rotation.x * d2r, roll = m.rotation.z * d2r;
        const double ch = std::cos(yaw), sh = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch), cb = std::cos(roll), sb = std::sin(roll);
        mat4 R;
        R.m[0] = (float)(ch * cb + sh * sp * sb); R.m[1] = (float)(sb * cp); R.m[2] = (float)(-sh * cb + ch * sp * sb);
        R.m[4] = (float)(-ch * sb + sh * sp * cb); R.m[5] = (float)(cb * cp); R.m[6] = (float)(sb * sh + ch * sp * cb);
        R.m[8] = (float)(sh * cp); R.m[9] = (float)(-sp); R.m[10] = (float)(ch * cp);
        mat4 T; T.m[12] = m.position.x; T.m[13] = m.position.y; T.m[14] = m.position.z;
        mat4 S; S.m[0] = m.scale.x; S.m[5] = m.scale.y; S.m[10] = m.scale.z;
        m.worldTransformMatrix = Mul(Mul(T, R), S);
    }
    static mat4 Mul(const mat4& a, const mat4& b) {          // (a*b)[col] = sum_k a[k] * b[col][k], accumulated in k order
        mat4 o;
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc = acc + a.m[k * 4 + r] * b.m[j * 4 + k];
            o.m[j * 4 + r] = acc;
        }
        return o;
    }
    // the vertex loop of Scene::AddNewMeshToScene / SceneManager (Scene.cpp:42-51, SceneManager.cpp:30-41):
    // position = (M * (p, 1)).xyz / w, normal = normalize((M * (n, 0)).xyz) — the model matrix, not its inverse transpose
    Vertex ToWorld(const Vertex& v) const {
        const float* M = worldTransformMatrix.m; Vertex o = v;
        const float px = (M[0] * v.position.x + M[4] * v.position.y) + (M[8] * v.position.z + M[12]);
        const float py = (M[1] * v.position.x + M[5] * v.position.y) + (M[9] * v.position.z + M[13]);
        const float pz = (M[2] * v.position.x + M[6] * v.position.y) + (M[10] * v.position.z + M[14]);
        const float pw = (M[3] * v.position.x + M[7] * v.position.y) + (M[11] * v.position.z + M[15]);
        o.position = vec3{px / pw, py / pw, pz / pw};
        const float nx = (M[0] * v.normal.x + M[4] * v.normal.y) + (M[8] * v.normal.z);
        const float ny = (M[1] * v.normal.x + M[5] * v.normal.y) + (M[9] * v.normal.z);
        const float nz = (M[2] * v.normal.x + M[6] * v.normal.y) + (M[10] * v.normal.z);
        const float inv = 1.0f / std::sqrt((nx * nx + ny * ny) + nz * nz);
        o.normal = vec3{nx * inv, ny * inv, nz * inv};
        return o;
    }
};
static_assert(sizeof(Vertex) == 32 && sizeof(Triangle) == 52 && sizeof(Material) == 44, "reference layouts");

struct Scene;
// SceneManager (Classes/Managers/SceneManager.{h,cpp}): the UI queues mesh / material edits (WalnutApp.cpp:662, :715) and
// PerformAllSceneUpdates applies them before the next frame and raises the renderer's scene-dirty flag (:776).  Both queues
// start with 20 default entries as the reference's do (SceneManager.h:25-26), so the first call always raises the flag; the
// emissive-triangle list is not refreshed here (the reference builds it only at load).  Acceleration structures are this
// build's own and are rebuilt by the library when the flag makes the facade upload the scene again.
struct SceneManager {
    struct MeshUpdateParam {
        MeshUpdateParam(bool t, bool m, uint32_t i) : meshTransformToBeUpdated(t), meshMatToBeUpdated(m), meshIndex(i) {}
        MeshUpdateParam() = default;
        bool meshTransformToBeUpdated = false, meshMatToBeUpdated = false; uint32_t meshIndex = 0xFFFFFFFFu;
    };
    std::vector<MeshUpdateParam> meshesToUpdate{20};
    std::vector<uint32_t> materialsToUpdate = std::vector<uint32_t>(20);
    template <class RendererT> void PerformAllSceneUpdates(Scene& scene, RendererT& renderer);
};

struct Scene {
    std::vector<Vertex> vertices;                  // object space (Scene::vertices)
    SceneManager sceneManager;
    std::vector<Vertex> worldVertices; std::vector<Triangle> triangles; std::vector<uint32_t> emissiveTriangles;
    std::vector<Mesh> meshes; std::vector<Material> materials; std::vector<Texture> textures;
    std::vector<TextureImage> textureStorage;   // owns the pixels of textures loaded through LoadTexture
    // Scene::LoadTexture / Texture::Texture(path) (Scene.cpp:225-260, Texture.cu:8-40): PNG -> ABGR words; returns the texture index or -1
    int LoadTexture(const std::string& path) {
        TextureImage img;
        if (!LoadTexturePNG(path, img)) return -1;
        textureStorage.push_back(std::move(img));
        textures.clear();                       // storage may have moved: rebuild the views
        for (TextureImage& t : textureStorage) { Texture v; v.pixels = t.pixels.data(); v.width = t.width; v.height = t.height; v.fileName = t.fileName; textures.push_back(v); }
        return (int)textures.size() - 1;
    }
    // Scene::AddNewMeshToScene (Scene.cpp:9-92): object-space vertices + transform -> world vertices, triangles, mesh record
    Mesh* AddNewMeshToScene(const std::vector<Vertex>& meshVertices, const std::vector<uint32_t>& indices, int materialIndex) {
        return AddNewMeshToScene(meshVertices, indices, vec3{0, 0, 0}, vec3{0, 0, 0}, vec3{1, 1, 1}, materialIndex);
    }
    Mesh* AddNewMeshToScene(const std::vector<Vertex>& meshVertices, const std::vector<uint32_t>& indices, const vec3& pos, const vec3& rotation,
                            const vec3& scale, int materialIndex) {
        Mesh mesh; mesh.vertexStart = (uint32_t)worldVertices.size(); mesh.vertexCount = (uint32_t)meshVertices.size();
        mesh.indexStart = (uint32_t)triangles.size() * 3u; mesh.indexCount = (uint32_t)indices.size(); mesh.materialIndex = materialIndex;
        mesh.position = pos; mesh.rotation = rotation; mesh.scale = scale; Mesh::UpdateWorldTransform(mesh);
        vertices.insert(vertices.end(), meshVertices.begin(), meshVertices.end());
        for (const Vertex& v : meshVertices) worldVertices.push_back(mesh.ToWorld(v));
        for (size_t i = 0; i + 2 < indices.size(); i += 3) {
            Triangle t; t.v0 = mesh.vertexStart + indices[i]; t.v1 = mesh.vertexStart + indices[i + 1]; t.v2 = mesh.vertexStart + indices[i + 2];
            t.materialIndex = materialIndex; triangles.push_back(t);
        }
        meshes.push_back(mesh); return &meshes.back();
    }
    void InitSceneEmissiveTriangles() {               // Scene.cpp:209-221
        emissiveTriangles.clear();
        for (uint32_t i = 0; i < triangles.size(); ++i) {
            const Material& m = materials[triangles[i].materialIndex];
            const float ex = m.emissionColor.x * m.emissionPower, ey = m.emissionColor.y * m.emissionPower, ez = m.emissionColor.z * m.emissionPower;
            if (ex * ex + ey * ey + ez * ez > 0.0f) emissiveTriangles.push_back(i);
        }
    }
};

template <class RendererT> void SceneManager::PerformAllSceneUpdates(Scene& scene, RendererT& renderer) {   // SceneManager.cpp:6-130
    if (!materialsToUpdate.empty()) { renderer.SetSceneToBeUpdatedFlag(true); renderer.NoteOtherSceneEdit(); }
    for (const MeshUpdateParam& u : meshesToUpdate) {
        if (!(u.meshTransformToBeUpdated || u.meshMatToBeUpdated)) continue;
        Mesh& mesh = scene.meshes[u.meshIndex];
        if (u.meshTransformToBeUpdated) {
            Mesh::UpdateWorldTransform(mesh);
            for (uint32_t i = 0; i < mesh.vertexCount; ++i) {
                const Vertex w = mesh.ToWorld(scene.vertices[mesh.vertexStart + i]);
                scene.worldVertices[mesh.vertexStart + i].position = w.position; scene.worldVertices[mesh.vertexStart + i].normal = w.normal;
            }
            renderer.SetSceneToBeUpdatedFlag(true);
            renderer.NoteMeshTransform(u.meshIndex, mesh.worldTransformMatrix.m);       // (the one line a maintainer adds to SceneManager.cpp:41)
        }
        if (u.meshMatToBeUpdated) {
            for (uint32_t t = mesh.indexStart / 3; t < mesh.indexStart / 3 + mesh.indexCount / 3; ++t) scene.triangles[t].materialIndex = mesh.materialIndex;
            renderer.SetSceneToBeUpdatedFlag(true); renderer.NoteOtherSceneEdit();
        }
    }
    meshesToUpdate.clear(); materialsToUpdate.clear();
}

// Camera.h:9-83 with the matrices filled by perspectiveFov / lookAt (glm RH, [-1,1] depth) and a general 4x4 inverse.
class Camera {
public:
    Camera(float verticalFOV, float nearClip, float farClip) : m_FOV(verticalFOV), m_Near(nearClip), m_Far(farClip) {}
    void OnResize(uint32_t w, uint32_t h) {
        if (w == m_W && h == m_H) return;
        m_W = w; m_H = h;
        const float rad = m_FOV * 3.14159265358979f / 180.0f, hh = std::cos(0.5f * rad) / std::sin(0.5f * rad), ww = hh * (float)h / (float)w;
        mat4 p; for (float& v : p.m) v = 0.0f;
        p.m[0] = ww; p.m[5] = hh; p.m[10] = -(m_Far + m_Near) / (m_Far - m_Near); p.m[11] = -1.0f; p.m[14] = -(2.0f * m_Far * m_Near) / (m_Far - m_Near);
        m_Projection = p; m_InverseProjection = Inverse(p);
    }
    void SetPosition(const vec3& p) { m_Position = p; UpdateCameraView(); }
    void SetDirection(const vec3& d) { m_Forward = d; UpdateCameraView(); }
    void SetPrevProjection(const mat4& m) { m_PrevProjection = m; }
    void SetPrevView(const mat4& m) { m_PrevView = m; }
    const mat4& GetProjection() const { return m_Projection; }
    const mat4& GetPrevProjection() const { return m_PrevProjection; }
    const mat4& GetInverseProjection() const { return m_InverseProjection; }
    const mat4& GetView() const { return m_View; }
    const mat4& GetPrevView() const { return m_PrevView; }
    const mat4& GetInverseView() const { return m_InverseView; }
    vec3& GetPosition() { return m_Position; }
    vec3& GetForwardDirection() { return m_Forward; }                      // (Camera.h: GetDirection; assigned without a view update by scenes that set the pose in two steps)
    uint32_t GetViewportWidth() const { return m_W; }
    uint32_t GetViewportHeight() const { return m_H; }
    // Camera::OnUpdate (Camera.cpp:18-94) with the right mouse button held; Walnut::Input is replaced by its values: `keys` = the
    // pressed keys out of "WSADQE", `mouseDelta` = cursor movement in pixels.  Recalculates the view and leaves the
    // previous-frame matrices alone (MainLayer::Render sets them after the frame, WalnutApp.cpp:908-909 == CommitFrame()).
    bool OnUpdate(float ts, const char* keys, vec2 mouseDelta) {
        const vec2 delta{mouseDelta.x * 0.002f, mouseDelta.y * 0.002f};
        auto has = [keys](char k) { for (const char* p = keys; p && *p; ++p) if (*p == k) return true; return false; };
        bool moved = false;
        const vec3 up{0, 1, 0}, f = m_Forward, right = Cross(f, up);
        const float speed = 5.0f;
        auto add = [&](vec3 d, float sgn) { m_Position = vec3{m_Position.x + sgn * (d.x * speed * ts), m_Position.y + sgn * (d.y * speed * ts), m_Position.z + sgn * (d.z * speed * ts)}; moved = true; };
        if (has('W')) add(f, 1.0f); else if (has('S')) add(f, -1.0f);
        if (has('A')) add(right, -1.0f); else if (has('D')) add(right, 1.0f);
        if (has('Q')) add(up, -1.0f); else if (has('E')) add(up, 1.0f);
        if (delta.x != 0.0f || delta.y != 0.0f) {
            const float pitch = delta.y * 0.3f, yaw = delta.x * 0.3f;
            float q[4]; QuatMul(AngleAxis(-pitch, right).q, AngleAxis(-yaw, up).q, q);
            const float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            for (float& c : q) c /= n;
            const vec3 qv{q[1], q[2], q[3]}, uv = Cross(qv, f), uuv = Cross(qv, uv);
            m_Forward = vec3{f.x + (uv.x * q[0] + uuv.x) * 2.0f, f.y + (uv.y * q[0] + uuv.y) * 2.0f, f.z + (uv.z * q[0] + uuv.z) * 2.0f};
            moved = true;
        }
        if (moved) RecalculateView();
        return moved;
    }
    void CommitFrame() { m_PrevProjection = m_Projection; m_PrevView = m_View; }
private:
    struct Quat { float q[4]; };                       // (w, x, y, z)
    static vec3 Cross(vec3 a, vec3 b) { return vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
    static Quat AngleAxis(float angle, vec3 axis) { const float h = angle * 0.5f, s = std::sin(h); return Quat{{std::cos(h), axis.x * s, axis.y * s, axis.z * s}}; }
    static void QuatMul(const float* a, const float* b, float* o) {
        o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3]; o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
        o[2] = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3]; o[3] = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1];
    }
    void RecalculateView() {                           // Camera.cpp:127-134 (lookAt towards position + forward, up = +y)
        auto nrm = [](vec3 v) { float l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); return vec3{v.x / l, v.y / l, v.z / l}; };
        auto dt = [](vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
        const vec3 f = nrm(m_Forward), s = nrm(Cross(f, vec3{0, 1, 0})), u = Cross(s, f);
        mat4 v;
        v.m[0] = s.x; v.m[4] = s.y; v.m[8] = s.z; v.m[1] = u.x; v.m[5] = u.y; v.m[9] = u.z; v.m[2] = -f.x; v.m[6] = -f.y; v.m[10] = -f.z;
        v.m[12] = -dt(s, m_Position); v.m[13] = -dt(u, m_Position); v.m[14] = dt(f, m_Position);
        m_View = v; m_InverseView = Inverse(v);
    }
    void UpdateCameraView() { RecalculateView(); m_PrevProjection = m_Projection; m_PrevView = m_View; }   // Camera.cpp:108-116
    static mat4 Inverse(const mat4& a) {               // Gauss–Jordan in double on the column-major array
        double m[4][8];
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { m[r][c] = a.m[c * 4 + r]; m[r][4 + c] = (r == c) ? 1.0 : 0.0; }
        for (int i = 0; i < 4; ++i) {
            int p = i; for (int r = i + 1; r < 4; ++r) if (std::fabs(m[r][i]) > std::fabs(m[p][i])) p = r;
            for (int c = 0; c < 8; ++c) std::swap(m[i][c], m[p][c]);
            const double d = m[i][i]; for (int c = 0; c < 8; ++c) m[i][c] /= d;
            for (int r = 0; r < 4; ++r) if (r != i) { const double f = m[r][i]; for (int c = 0; c < 8; ++c) m[r][c] -= f * m[i][c]; }
        }
        mat4 o; for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) o.m[c * 4 + r] = (float)m[r][4 + c];
        return o;
    }
    mat4 m_Projection, m_View, m_PrevProjection, m_PrevView, m_InverseProjection, m_InverseView;
    float m_FOV, m_Near, m_Far; vec3 m_Position{0, 0, 6}, m_Forward{0, 0, -1}; uint32_t m_W = 0, m_H = 0;
};

}  // namespace fyprt_host
