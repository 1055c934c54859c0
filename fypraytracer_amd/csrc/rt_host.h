// Host-side data contracts of libfyprt: the acceleration-structure layout that is uploaded
// to HBM and the builders' entry points.
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/fyprt.h"

namespace rth {

// 64-byte node = four 16-byte quads, both child boxes in the parent so one node fetch
// (4 x global_load_dwordx4) decides both children.
//   q0 = lo0.x lo0.y lo0.z hi0.x | q1 = hi0.y hi0.z lo1.x lo1.y | q2 = lo1.z hi1.x hi1.y hi1.z | q3 = child0 child1 pad pad
// child >= 0: inner node index.  child < 0: leaf, ~child = (firstTri << 2) | (count - 1), count 1..4.
struct Node { float lo0[3], hi0[3], lo1[3], hi1[3]; int32_t child0, child1, pad[2]; };
// 48-byte leaf triangle = three quads: v0.xyz e1.x | e1.yz e2.xy | e2.z tri pad pad  (e1 = v1 - v0, e2 = v2 - v0)
struct Tri { float v0[3], e1[3], e2[3]; uint32_t tri, pad[2]; };
static_assert(sizeof(Node) == 64 && sizeof(Tri) == 48, "layout");

struct SceneBVH {
    std::vector<Node> nodes;   // TLAS nodes first, then each mesh's BLAS
    std::vector<Tri> tris;     // leaf order
    int32_t rootRef = 0;       // inner index or leaf code
    uint32_t maxDepth = 0;     // deepest leaf (root = depth 0) -> traversal stack bound
    uint32_t tlasNodes = 0;
};

// Two-level build: one SAH BLAS per mesh (world-space triangles, <= 4 per leaf) and a SAH TLAS
// over the mesh bounds whose leaves are the BLAS roots.  Replaces BVH::ConstructBVH_SAH
// (BVH.cpp:65-81) + Mesh::CreateBVHnodesFromMeshTriangles / Scene::CreateBVHnodesFromSceneMeshes.
void BuildSceneBVH(const fyprt_vertex* verts, const uint8_t* tris, uint32_t triStride, const fyprt_mesh* meshes,
                   uint32_t meshCount, SceneBVH& out);

struct LightTrees {
    std::vector<fyprt_lighttree_node> tlas; uint32_t tlasRoot = ~0u;
    std::vector<fyprt_lighttree_node> blas; std::vector<uint32_t> first, count, root;
};
// Restatement of LightTree::ConstructLightTree (LightTree.cpp:4-340) + the leaf producers
// (Mesh.cpp:176-207, Scene.cpp:160-186), in the flat node format of fyprt.h.
void BuildLightTrees(const fyprt_vertex* verts, const uint8_t* tris, uint32_t triStride, const fyprt_mesh* meshes,
                     uint32_t meshCount, const fyprt_material* mats, LightTrees& out);

}  // namespace rth
