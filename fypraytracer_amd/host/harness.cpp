// Headless harness: what MainLayer (WalnutApp.cpp:43-520, :878-910) does minus the GUI — build a
// scene through the Scene API, drive Renderer::OnResize / Render per frame (randSeed++ as
// MainLayer::OnUpdate does, WalnutApp.cpp:532; prev matrices committed as :908-909), print the
// reference's on-screen statistics (frame time, accumulated frames) and write the frame as a
// bottom-up 24-bit BMP (MisUtils::SaveABGRToBMP, MisUtils.cpp:13-95).
//   usage: harness [technique 0-8] [frames] [width] [height] [out.bmp]
#include <chrono>
#include <cstdlib>
#include <fstream>
#include "HostTypes.h"
#include "Renderer.h"
using namespace fyprt_host;

static void quad(Scene& s, vec3 a, vec3 b, vec3 c, vec3 d, vec3 n, int mat) {
    std::vector<Vertex> v = {{a, n, {0, 0}}, {b, n, {1, 0}}, {c, n, {1, 1}}, {d, n, {0, 1}}};
    s.AddNewMeshToScene(v, {0, 1, 2, 0, 2, 3}, mat);
}
static void saveBmp(const char* path, const uint32_t* abgr, uint32_t w, uint32_t h) {
    const uint32_t row = (w * 3 + 3) & ~3u, size = 54 + row * h;
    std::vector<uint8_t> f(size, 0);
    f[0] = 'B'; f[1] = 'M'; std::memcpy(&f[2], &size, 4); uint32_t off = 54, hs = 40; std::memcpy(&f[10], &off, 4); std::memcpy(&f[14], &hs, 4);
    std::memcpy(&f[18], &w, 4); std::memcpy(&f[22], &h, 4); uint16_t planes = 1, bpp = 24; std::memcpy(&f[26], &planes, 2); std::memcpy(&f[28], &bpp, 2);
    for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) {      // row 0 of the render = bottom of the picture = first BMP row
        const uint32_t p = abgr[y * w + x]; uint8_t* q = &f[54 + y * row + x * 3];
        q[0] = (p >> 16) & 0xFF; q[1] = (p >> 8) & 0xFF; q[2] = p & 0xFF;
    }
    std::ofstream(path, std::ios::binary).write((const char*)f.data(), f.size());
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
    for (int f = 0; f < frames; ++f) {
        s.randSeed++;                                                                       // WalnutApp.cpp:532
        const auto t0 = std::chrono::steady_clock::now();
        renderer.Render(scene, camera);
        total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        camera.SetPrevProjection(camera.GetProjection()); camera.SetPrevView(camera.GetView()); // WalnutApp.cpp:908-909
    }
    std::printf("Resolution : %ux%u\nTriangles : %zu\nAvg frame time : %.3fms (kernels %.3fms)\nAccumulated frames : %u\n",
                W, H, scene.triangles.size(), total / frames, renderer.GetLastFrameStats().kernel_ms, renderer.GetCurrentFrameIndex() - 1);
    if (argc > 5) saveBmp(argv[5], renderer.GetRenderImageDataPtr(), W, H);
    return 0;
}
