// Command-line face of the host scene API for tests/test_host_scene.py (host-only, no GPU): builds one mesh from a fixed
// vertex list with the transform given on the command line, optionally moves it through the SceneManager, and prints the
// world vertices as hex floats so the Python mirror (scene.py) can be compared bit for bit.
//   scene_check px py pz rx ry rz sx sy sz [px2 py2 pz2 rx2 ry2 rz2]
#include <cstdlib>
#include "HostTypes.h"
using namespace fyprt_host;
struct FlagOnly { bool dirty = false; void SetSceneToBeUpdatedFlag(bool f) { dirty = f; } void NoteMeshTransform(uint32_t, const float*) {} void NoteOtherSceneEdit() {} };
int main(int argc, char** argv) {
    if (argc != 10 && argc != 16) return 2;
    float a[15] = {0}; for (int i = 1; i < argc; ++i) a[i - 1] = (float)std::atof(argv[i]);
    std::vector<Vertex> v;
    for (int i = 0; i < 7; ++i) {
        Vertex x; x.position = vec3{0.25f * i - 0.7f, 0.1f * i * i - 0.3f, 1.0f - 0.37f * i};
        x.normal = vec3{0.3f + 0.1f * i, -0.5f + 0.2f * i, 0.8f - 0.15f * i}; x.uv = vec2{0.1f * i, 1.0f - 0.1f * i}; v.push_back(x);
    }
    Scene scene; Material m; scene.materials = {m, m};
    scene.AddNewMeshToScene(v, {0, 1, 2, 2, 3, 4, 4, 5, 6}, vec3{a[0], a[1], a[2]}, vec3{a[3], a[4], a[5]}, vec3{a[6], a[7], a[8]}, 0);
    FlagOnly r;
    scene.sceneManager.PerformAllSceneUpdates(scene, r);
    std::printf("first_call_dirty %d\n", (int)r.dirty);
    r.dirty = false; scene.sceneManager.PerformAllSceneUpdates(scene, r);
    std::printf("second_call_dirty %d\n", (int)r.dirty);
    if (argc == 16) {
        Mesh& me = scene.meshes[0]; me.position = vec3{a[9], a[10], a[11]}; me.rotation = vec3{a[12], a[13], a[14]}; me.materialIndex = 1;
        scene.sceneManager.meshesToUpdate.emplace_back(true, true, 0u);
        scene.sceneManager.PerformAllSceneUpdates(scene, r);
        std::printf("moved_dirty %d material %d\n", (int)r.dirty, scene.triangles[2].materialIndex);
    }
    for (const Vertex& w : scene.worldVertices) std::printf("%a %a %a %a %a %a\n", w.position.x, w.position.y, w.position.z, w.normal.x, w.normal.y, w.normal.z);
    return 0;
}
