// Headless harness: what MainLayer (WalnutApp.cpp:43-520, :878-910) does minus the GUI — build a
// scene through the Scene API, drive Renderer::OnResize / Render per frame (randSeed++ as
// MainLayer::OnUpdate does, WalnutApp.cpp:532; prev matrices committed as :908-909), print the
// reference's on-screen statistics (frame time, accumulated frames) and write the frame as a
// bottom-up 24-bit BMP (MisUtils::SaveABGRToBMP, MisUtils.cpp:13-95).
// With a reference image it also prints the benchmark record name MainLayer::SaveBenchmarkResults builds
// (WalnutApp.cpp:833-875: frame times, technique, its parameters, MSE, PSNR).
//   usage: harness [technique 0-8] [frames] [width] [height] [out.bmp] [reference.bmp]
#include <chrono>
#include <cstdlib>
#include <fstream>
#include "HostTypes.h"
#include "MisUtils.h"
#include "Renderer.h"
using namespace fyprt_host;

static void quad(Scene& s, vec3 a, vec3 b, vec3 c, vec3 d, vec3 n, int mat) {
    std::vector<Vertex> v = {{a, n, {0, 0}}, {b, n, {1, 0}}, {c, n, {1, 1}}, {d, n, {0, 1}}};
    s.AddNewMeshToScene(v, {0, 1, 2, 0, 2, 3}, mat);
}
int main(int argc, char** argv) {
    const int tech = argc > 1 ? std::atoi(argv[1]) : RESTIR_DI, frames = argc > 2 ? std::atoi(argv[2]) : 16;
    const uint32_t W = argc > 3 ? std::atoi(argv[3]) : 512, H = argc > 4 ? std::atoi(argv[4]) : 512;
    Scene scene;
    Material white; white.albedo = {1, 1, 1}; Material red; red.albedo = {1, 0, 0}; Material green; green.albedo = {0, 1, 0};
    Material light; light.albedo = {1, 1, 1}; light.emissionColor = {1, 1, 1}; light.emissionPower = 40.0f;   // WalnutApp.cpp:56-59
    scene.materials = {white, red, green, light};
    quad(scene, {-1, -1, 1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1}, {0, 1, 0}, 0);
    quad(scene, {-1, 1, -1}, {1, 1, -1}, {1, 1, 1}, {-1, 1, 1}, {0, -1, 0}, 0);
    quad(scene, {-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {0, 0, 1}, 0);
    quad(scene, {-1, -1, 1}, {-1, -1, -1}, {-1, 1, -1}, {-1, 1, 1}, {1, 0, 0}, 1);
    quad(scene, {1, -1, -1}, {1, -1, 1}, {1, 1, 1}, {1, 1, -1}, {-1, 0, 0}, 2);
    quad(scene, {-0.25f, 0.999f, -0.25f}, {0.25f, 0.999f, -0.25f}, {0.25f, 0.999f, 0.25f}, {-0.25f, 0.999f, 0.25f}, {0, -1, 0}, 3);
    scene.InitSceneEmissiveTriangles();
    Camera camera(45.0f, 0.1f, 100.0f);
    Renderer renderer(0);
    RenderingSettings& s = renderer.GetSettings();
    s.currentSamplingTechnique = (SamplingTechniqueEnum)tech; s.lightBounces = 4; s.skyColor[0] = s.skyColor[1] = s.skyColor[2] = 0.0f;
    s.useTemporalReuse = s.useSpatialReuse = true;
    renderer.OnResize(W, H); camera.OnResize(W, H); camera.SetPosition({0, 0, 3.4f});
    double total = 0.0;
    scene.sceneManager.PerformAllSceneUpdates(scene, renderer);   // MainLayer::OnUpdate calls it every frame (WalnutApp.cpp:776); the first call drains the 20 default queue entries
    for (int f = 0; f < frames; ++f) {
        s.randSeed++;                                                                       // WalnutApp.cpp:532
        const auto t0 = std::chrono::steady_clock::now();
        renderer.Render(scene, camera);
        total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        camera.SetPrevProjection(camera.GetProjection()); camera.SetPrevView(camera.GetView()); // WalnutApp.cpp:908-909
    }
    {   // one transform edit through the SceneManager (WalnutApp.cpp:620-662, :776): the facade refits on the device
        Mesh& lightMesh = scene.meshes.back();
        lightMesh.position = vec3{0.05f, 0.0f, 0.0f};
        scene.sceneManager.meshesToUpdate.emplace_back(true, false, (uint32_t)scene.meshes.size() - 1u);
        scene.sceneManager.PerformAllSceneUpdates(scene, renderer);
        s.randSeed++;
        renderer.Render(scene, camera);
    }
    std::printf("Scene uploads : %u, device refits : %u (of them by matrix alone : %u)\n", renderer.GetSceneUploadCount(), renderer.GetSceneRefitCount(), renderer.GetTransformUpdateCount());
    std::printf("Resolution : %ux%u\nTriangles : %zu\nAvg frame time : %.3fms (kernels %.3fms)\nAccumulated frames : %u\n",
                W, H, scene.triangles.size(), total / frames, renderer.GetLastFrameStats().kernel_ms, renderer.GetCurrentFrameIndex() - 1);
    if (argc > 5 && !MisUtils::SaveABGRToBMP(argv[5], renderer.GetRenderImageDataPtr(), (int)W, (int)H)) { std::fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
    if (argc > 6) {
        std::vector<uint32_t> ref; uint32_t rw = 0, rh = 0;
        if (!MisUtils::LoadBMPToABGR(argv[6], ref, rw, rh) || rw != W || rh != H) { std::fprintf(stderr, "cannot use reference image %s\n", argv[6]); return 1; }
        const double mse = MisUtils::ComputeMSE(ref.data(), renderer.GetRenderImageDataPtr(), W, H);
        std::printf("Record : %s\n", MisUtils::BenchmarkRecordName(s, (float)(total / frames), (float)total, true, mse, MisUtils::ComputePSNR(mse)).c_str());
    } else {
        std::printf("Record : %s\n", MisUtils::BenchmarkRecordName(s, (float)(total / frames), (float)total).c_str());
    }
    return 0;
}
