// Host-side data contracts of libfyprt: the acceleration-structure layout that is uploaded
// to HBM and the builders' entry points.
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/fyprt.h"

namespace rth {

// 64-byte 4-wide node = four 16-byte quads; the child boxes are quantised to 8 bits per plane on a power-of-two grid
// anchored at the node's own box, so ONE node fetch (4 x global_load_dwordx4) decides up to four children:
//   q0 = origin.x origin.y origin.z (ex | ey << 8 | ez << 16 | meta << 24)      grid step on axis a = 2^(e_a - 127)
//        meta = count | levels << 3: children 0..count-1 are valid (count 2..4); levels = wide levels of this subtree (1 = leaves below)
//   q1 = child[0..3]
//   q2 = qlo.x[0..3] qlo.y[0..3] qlo.z[0..3] qhi.x[0..3]                        one byte per child, child i = byte i
//   q3 = qhi.y[0..3] qhi.z[0..3] pad pad
// plane = origin_a + q * 2^(e_a - 127); lo planes are rounded down and hi planes up (with 1/16 step of slack), so the
// quantised box always contains the exact child box.  Unused slots hold qlo = 255, qhi = 0 on every axis.
// child >= 0: inner node index (< 2^26).  child < 0: leaf, ~child = (firstTri << 2) | (count - 1), count 1..4.
struct Node { float origin[3]; uint8_t ex[3]; uint8_t meta; int32_t child[4]; uint8_t qlo[3][4]; uint8_t qhi[3][4]; uint32_t pad[2]; };
// 48-byte leaf triangle = three quads: v0.xyz e1.x | e1.yz e2.xy | e2.z tri pad pad  (e1 = v1 - v0, e2 = v2 - v0)
struct Tri { float v0[3], e1[3], e2[3]; uint32_t tri, pad[2]; };
static_assert(sizeof(Node) == 64 && sizeof(Tri) == 48, "layout");

struct SceneBVH {
    std::vector<Node> nodes;   // pre-order: the top of the tree is contiguous
    std::vector<Tri> tris;     // leaf order
    int32_t rootRef = 0;       // inner index or leaf code
    uint32_t maxDepth = 0;     // deepest leaf of the binary SAH tree the wide tree is collapsed from (root = depth 0)
    uint32_t levels = 0;       // wide levels of the tree (<= kStackBudget, see node_step's stack rule)
    uint32_t binaryNodes = 0;  // inner nodes of the binary tree before the collapse
};

constexpr uint32_t kStackBudget = 31;   // kernels: 32-entry LDS stack, one entry is the exit sentinel
// Child references of inner nodes have two forms.  HOST form (SceneBVH, fyprt_export_bvh, the oracle's twin): the node's index.  DEVICE form
// (the node array in HBM, DevScene::rootRef, the traversal stack): the node's BYTE OFFSET, index << 6 — a visit then forms its address with
// one v_and and a 32-bit offset on the array's base instead of a resume-entry decode and a 64-bit shift + add (rt_device.h: node_step; r03).
// Leaf codes (negative) are the same in both forms; offsets stay below 2^30 (bit 30 marks a resume entry), so a tree has < 2^24 nodes.
constexpr int kNodeRefShift = 6;
constexpr uint32_t kMaxNodes = 1u << 24;
inline int32_t device_ref(int32_t hostRef) { return hostRef >= 0 ? (int32_t)((uint32_t)hostRef << kNodeRefShift) : hostRef; }
inline int32_t host_ref(int32_t deviceRef) { return deviceRef >= 0 ? (deviceRef >> kNodeRefShift) : deviceRef; }
inline void nodes_to_device_form(Node* n, size_t count) { for (size_t i = 0; i < count; ++i) for (int k = 0; k < 4; ++k) n[i].child[k] = device_ref(n[i].child[k]); }
inline void nodes_to_host_form(Node* n, size_t count) { for (size_t i = 0; i < count; ++i) for (int k = 0; k < 4; ++k) n[i].child[k] = host_ref(n[i].child[k]); }
// Two-level build: one SAH BLAS per mesh (world-space triangles, <= 4 per leaf) and a SAH TLAS
// over the mesh bounds whose leaves are the BLAS roots, merged into one binary tree and then collapsed
// into 4-wide nodes with quantised child boxes.  Replaces BVH::ConstructBVH_SAH
// (BVH.cpp:65-81) + Mesh::CreateBVHnodesFromMeshTriangles / Scene::CreateBVHnodesFromSceneMeshes.
void BuildSceneBVH(const fyprt_vertex* verts, const uint8_t* tris, uint32_t triStride, const fyprt_mesh* meshes,
                   uint32_t meshCount, SceneBVH& out);

struct LightTrees {
    std::vector<fyprt_lighttree_node> tlas; uint32_t tlasRoot = ~0u;
    std::vector<fyprt_lighttree_node> blas; std::vector<uint32_t> first, count, root;
};
// Restatement of LightTree::ConstructLightTree (LightTree.cpp:4-340) + the leaf producers
// (Mesh.cpp:176-207, Scene.cpp:160-186), in the flat node format of fyprt.h.
// `touched` (one byte per mesh) != nullptr: `out` already holds the trees of this topology and only the touched meshes' trees are
// rebuilt (vertices of the other meshes are not read), then the TLAS; the result equals a full build.
void BuildLightTrees(const fyprt_vertex* verts, const uint8_t* tris, uint32_t triStride, const fyprt_mesh* meshes,
                     uint32_t meshCount, const fyprt_material* mats, LightTrees& out, const uint8_t* touched = nullptr);

}  // namespace rth
