// C++ host facade over the C ABI (include/fyprt.h) with the public surface of the reference's
// `Renderer` (FYPRayTracer/src/Classes/Core/Renderer.h:41-56), so the call sites in
// MainLayer::Render (WalnutApp.cpp:878-910) and SceneManager (SceneManager.cpp:14,65,81) keep
// their shape:  OnResize / Render(scene, camera) / ResetFrameIndex / GetSettings /
// GetCurrentFrameIndex / GetRenderImageDataPtr / Resize*Buffers / FreeDynamicallyAllocatedMemory /
// SetSceneToBeUpdatedFlag.  `GetFinalRenderImage()` (a Walnut::Image, i.e. a Vulkan upload) becomes
// an optional presenter callback — Walnut/Vulkan is outside this path.
//
// Templated on the caller's Scene / Camera types: anything exposing the reference's member names
// (`worldVertices`, `triangles`, `materials`, `meshes[i].{indexStart,indexCount,materialIndex}`,
// `textures[i].{pixels,width,height}`; camera getters `GetProjection()` ... `GetPosition()`,
// `GetViewportWidth/Height()`) works — the reference's own classes do, and so do the glm-free
// stand-ins in HostTypes.h used by the headless harness.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>
#include "../../include/fyprt.h"

namespace fyprt_host {

// RenderingSettings.h:5-22, same field names / defaults; layout-compatible with fyprt_settings.
enum SamplingTechniqueEnum { BRUTE_FORCE, UNIFORM_SAMPLING, COSINE_WEIGHTED_SAMPLING, GGX_SAMPLING, BRDF_SAMPLING,
                             LIGHT_SOURCE_SAMPLING, NEE, RESTIR_DI, RESTIR_GI, SamplingTechniqueEnum_COUNT };
struct RenderingSettings {
    bool toAccumulate = true; int lightBounces = 1; int sampleCount = 1; float skyColor[3] = {1, 1, 1};
    SamplingTechniqueEnum currentSamplingTechnique = BRUTE_FORCE; int lightCandidateCount = 4; uint32_t randSeed = 1;
    bool useTemporalReuse = false; bool useSpatialReuse = false; int temporalHistoryLimit = 2; int spatialNeighborNum = 5; int spatialNeighborRadius = 30;
};
static_assert(sizeof(RenderingSettings) == sizeof(fyprt_settings), "RenderingSettings must stay 52 bytes");

class Renderer {
public:
    explicit Renderer(int device = 0) { report(fyprt_create(device, &m_Ctx), "fyprt_create"); }
    ~Renderer() { FreeDynamicallyAllocatedMemory(); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void OnResize(uint32_t width, uint32_t height) {                       // Renderer.cpp:5-41
        if (width == m_Width && height == m_Height) return;
        if (report(fyprt_resize(m_Ctx, width, height), "fyprt_resize")) return;
        m_Width = width; m_Height = height;
        m_RenderImageData.assign((size_t)width * height, 0u);
        m_AccumulationData.assign((size_t)width * height * 4, 0.0f);
    }

    template <class SceneT, class CameraT> void Render(SceneT& scene, CameraT& camera) {   // Renderer.cu:13-284
        if (isSceneUpdated) {                                                                 // :61-69
            // only vertices moved since the last upload (a transform edit): refit on the device instead of a full rebuild
            const uint64_t sig = TopologySignature(scene);
            // (a) nothing but mesh transforms changed and the SceneManager told us which (NoteMeshTransform): 64 bytes per mesh go to the
            //     device, which recomputes the world vertices itself;  (b) same topology: every world vertex is uploaded, refit on the device
            if (m_HaveScene && sig == m_TopologySig && !m_PendingTransforms.empty() && !m_OtherEdits &&
                fyprt_update_transforms(m_Ctx, m_PendingMeshes.data(), m_PendingTransforms.data(), (uint32_t)m_PendingMeshes.size()) == FYPRT_OK) {
                ++m_Refits; ++m_TransformUpdates;
            } else if (m_HaveScene && sig == m_TopologySig &&
                fyprt_update_vertices(m_Ctx, reinterpret_cast<const fyprt_vertex*>(scene.worldVertices.data()), (uint32_t)scene.worldVertices.size()) == FYPRT_OK) {
                ++m_Refits;
            } else if (UploadScene(scene)) {                 // remember the topology only of a scene the library really holds
                m_TopologySig = sig; m_HaveScene = true; ++m_Uploads;
            } else {
                m_HaveScene = false;
            }
            isSceneUpdated = false; m_PendingMeshes.clear(); m_PendingTransforms.clear(); m_OtherEdits = false;
        }
        fyprt_camera_desc c{};                                                                // :70 CameraToGPU
        std::memcpy(c.projection, &camera.GetProjection(), 64); std::memcpy(c.view, &camera.GetView(), 64);
        std::memcpy(c.prev_projection, &camera.GetPrevProjection(), 64); std::memcpy(c.prev_view, &camera.GetPrevView(), 64);
        std::memcpy(c.inverse_projection, &camera.GetInverseProjection(), 64); std::memcpy(c.inverse_view, &camera.GetInverseView(), 64);
        std::memcpy(c.position, &camera.GetPosition(), 12);
        c.viewport_width = camera.GetViewportWidth(); c.viewport_height = camera.GetViewportHeight();
        if (report(fyprt_set_camera(m_Ctx, &c), "fyprt_set_camera")) return;
        fyprt_settings s; std::memcpy(&s, &m_Settings, sizeof s);
        if (report(fyprt_render(m_Ctx, &s, &m_LastStats), "fyprt_render")) return;           // :87-237
        report(fyprt_readback(m_Ctx, m_RenderImageData.data(), m_AccumulationData.data()), "fyprt_readback");   // :244-250
        if (m_Present) m_Present(m_RenderImageData.data(), m_Width, m_Height);                // :256 SetData
    }

    void ResetFrameIndex() { fyprt_reset_frame_index(m_Ctx); }
    RenderingSettings& GetSettings() { return m_Settings; }
    uint32_t GetCurrentFrameIndex() const { return fyprt_frame_index(m_Ctx); }
    uint32_t* GetRenderImageDataPtr() const { return const_cast<uint32_t*>(m_RenderImageData.data()); }
    const float* GetAccumulationDataPtr() const { return m_AccumulationData.data(); }
    // The four Resize*Buffers calls of the reference (Renderer.cu:286-419) re-zero the ReSTIR state.
    void ResizeReservoirs(uint32_t w, uint32_t h) { Rezero(w, h); }
    void ResizeDepthBuffers(uint32_t w, uint32_t h) { Rezero(w, h); }
    void ResizeNormalBuffers(uint32_t w, uint32_t h) { Rezero(w, h); }
    void ResizePrimaryHitPayloadBuffers(uint32_t w, uint32_t h) { Rezero(w, h); }
    void FreeDynamicallyAllocatedMemory() { if (m_Ctx) { fyprt_destroy(m_Ctx); m_Ctx = nullptr; } }
    void SetSceneToBeUpdatedFlag(bool flag) { isSceneUpdated = flag; }
    // Called by SceneManager::PerformAllSceneUpdates next to SetSceneToBeUpdatedFlag for a mesh whose transform changed (its
    // Mesh::worldTransformMatrix, column-major) — lets Render() send the matrix instead of the mesh's vertices.  Any other edit
    // (material change, ...) is reported with NoteOtherSceneEdit() and takes the general path.
    void NoteMeshTransform(uint32_t meshIndex, const float* matrix16) { m_PendingMeshes.push_back(meshIndex); m_PendingTransforms.insert(m_PendingTransforms.end(), matrix16, matrix16 + 16); }
    void NoteOtherSceneEdit() { m_OtherEdits = true; }
    uint32_t GetTransformUpdateCount() const { return m_TransformUpdates; }
    void SetPresenter(std::function<void(const uint32_t*, uint32_t, uint32_t)> p) { m_Present = std::move(p); }
    const fyprt_frame_stats& GetLastFrameStats() const { return m_LastStats; }
    fyprt_context* GetContext() const { return m_Ctx; }
    uint32_t GetSceneUploadCount() const { return m_Uploads; }
    uint32_t GetSceneRefitCount() const { return m_Refits; }

private:
    // FNV-1a over everything of the scene except vertex positions / normals: triangles (indices + material), meshes, materials,
    // emissive list, texture identities, vertex count
    template <class SceneT> static uint64_t TopologySignature(const SceneT& scene) {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&h](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
        const uint64_t counts[5] = {scene.worldVertices.size(), scene.triangles.size(), scene.meshes.size(), scene.materials.size(), scene.textures.size()};
        mix(counts, sizeof counts);
        for (const auto& t : scene.triangles) { const uint32_t v[4] = {t.v0, t.v1, t.v2, (uint32_t)t.materialIndex}; mix(v, sizeof v); }
        for (const auto& m : scene.meshes) { const uint32_t v[3] = {m.indexStart, m.indexCount, (uint32_t)m.materialIndex}; mix(v, sizeof v); }
        if (!scene.materials.empty()) mix(scene.materials.data(), scene.materials.size() * sizeof(scene.materials[0]));
        if (!scene.emissiveTriangles.empty()) mix(scene.emissiveTriangles.data(), scene.emissiveTriangles.size() * 4);
        for (const auto& t : scene.textures) { const uint64_t v[3] = {(uint64_t)(uintptr_t)t.pixels, t.width, t.height}; mix(v, sizeof v); }
        return h;
    }
    template <class SceneT> bool UploadScene(SceneT& scene) {                                 // SceneToGPU, Scene_GPU.cpp:6-81
        fyprt_scene_desc d{};
        d.vertices = reinterpret_cast<const fyprt_vertex*>(scene.worldVertices.data()); d.vertex_count = (uint32_t)scene.worldVertices.size();
        d.triangles = scene.triangles.data(); d.triangle_count = (uint32_t)scene.triangles.size();
        d.triangle_stride = (uint32_t)sizeof(scene.triangles[0]);
        d.materials = reinterpret_cast<const fyprt_material*>(scene.materials.data()); d.material_count = (uint32_t)scene.materials.size();
        std::vector<fyprt_mesh> meshes(scene.meshes.size());
        for (size_t i = 0; i < meshes.size(); ++i)
            meshes[i] = fyprt_mesh{scene.meshes[i].indexStart / 3u, scene.meshes[i].indexCount / 3u, scene.meshes[i].materialIndex};
        d.meshes = meshes.data(); d.mesh_count = (uint32_t)meshes.size();
        std::vector<fyprt_texture> tex(scene.textures.size());
        for (size_t i = 0; i < tex.size(); ++i) tex[i] = fyprt_texture{scene.textures[i].pixels, scene.textures[i].width, scene.textures[i].height};
        d.textures = tex.data(); d.texture_count = (uint32_t)tex.size();
        d.emissive_triangles = scene.emissiveTriangles.empty() ? nullptr : scene.emissiveTriangles.data();
        d.emissive_count = (uint32_t)scene.emissiveTriangles.size();
        d.light_trees = nullptr;                      // the library builds them (LightTree.cpp restated)
        if (report(fyprt_upload_scene(m_Ctx, &d), "fyprt_upload_scene")) return false;
        // object-space vertices + mesh vertex ranges: what fyprt_update_transforms needs (Scene::vertices, Mesh::vertexStart / vertexCount)
        if (scene.vertices.size() == scene.worldVertices.size()) {
            std::vector<uint32_t> first(scene.meshes.size() + 1, (uint32_t)scene.vertices.size());
            for (size_t i = 0; i < scene.meshes.size(); ++i) first[i] = scene.meshes[i].vertexStart;
            fyprt_set_object_vertices(m_Ctx, reinterpret_cast<const fyprt_vertex*>(scene.vertices.data()), (uint32_t)scene.vertices.size(), first.data());
        }
        return true;
    }
    void Rezero(uint32_t w, uint32_t h) { m_Width = m_Height = 0; OnResize(w, h); }
    bool report(int rc, const char* what) {          // the reference prints and keeps going (Renderer.cu:29-47)
        if (rc == FYPRT_OK) return false;
        std::fprintf(stderr, "%s error: %s\n", what, fyprt_last_error(m_Ctx));
        return true;
    }
    RenderingSettings m_Settings;
    fyprt_context* m_Ctx = nullptr;
    uint32_t m_Width = 0, m_Height = 0;
    std::vector<uint32_t> m_RenderImageData;
    std::vector<float> m_AccumulationData;
    bool isSceneUpdated = true;
    bool m_HaveScene = false; uint64_t m_TopologySig = 0; uint32_t m_Uploads = 0, m_Refits = 0, m_TransformUpdates = 0;
    std::vector<uint32_t> m_PendingMeshes; std::vector<float> m_PendingTransforms; bool m_OtherEdits = false;
    std::function<void(const uint32_t*, uint32_t, uint32_t)> m_Present;
    fyprt_frame_stats m_LastStats{};
};

}  // namespace fyprt_host
