// Host-side construction of the traversal layout the extend kernel walks.
//
// Input is the interchange form of include/rrt_amd.h: sphere planes, material planes and the BVH
// nodes exactly as the reference's Bvh::build_r pushed them (src/acceleration.rs:150-217).
// Output is a 64-byte-per-node array in which every node carries, for each of the 8 ray-direction
// sign octants, the node that the reference's stack would pop next once the node's subtree is
// exhausted (src/acceleration.rs:236,253-267).  With those links the traversal needs no stack and
// visits nodes in exactly the reference's order.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rrt_amd.h"

namespace rrt_amd {

constexpr uint32_t kRefLeafBit = 0x80000000u;   // a node reference that points at a leaf
constexpr uint32_t kRefEnd = 0x7FFFFFFFu;       // "nothing left to visit"
constexpr uint32_t kRefIndexMask = 0x1FFFFFFFu; // node index inside a reference
constexpr uint32_t kNodeWords = 16;             // 64 bytes
constexpr uint32_t kLeafPrimMask = 0x3FFFFFFFu; // leaf word 4: sphere index | material kind << 30
constexpr uint32_t kLeafKindShift = 30;

// Compact layout for scenes small enough to live in a workgroup's LDS.  A 16-bit reference is the node's offset in
// 16-byte units (at most 4095 inside the 64 KB budget) with bit 15 set for a leaf.
constexpr uint32_t kRef16LeafBit = 0x8000u;
constexpr uint32_t kRef16End = 0x7FFFu;
constexpr uint32_t kSmallWordsStatic = 12;   // 48 bytes per node
constexpr uint32_t kSmallWordsMoving = 16;   // 64 bytes per node: + centre1 - centre0
constexpr uint32_t kSmallLdsBudget = 64u * 1024u;

// Per-sphere info word: material index | kind << 28 | moving << 31
constexpr uint32_t kInfoMatMask = 0x0FFFFFFFu;
constexpr uint32_t kInfoKindShift = 28;
constexpr uint32_t kInfoMovingBit = 0x80000000u;

// Scene flags
constexpr uint32_t kSceneHasMoving = 1u;
constexpr uint32_t kSceneUniformTimes = 2u;

struct Float4 { float x, y, z, w; };

struct SceneLayout {
    // Node words.  Inner: [0..5] box as (min, max) per axis: x, y, z, [6] right ref | axis << 29, [7] left ref,
    // [8..15] exit link per octant.  Leaf: [0..2] centre at time0, [3] radius^2, [4] sphere index | material kind << 30,
    // [5] bit 0 = moving, [8..15] exit link per octant.
    std::vector<uint32_t> nodes;
    uint32_t n_nodes = 0;
    uint32_t root_ref = kRefEnd;
    uint32_t max_leaf_depth = 0;

    // Compact node words (empty when the scene does not qualify).  [0..5] inner: (min, max) per axis; leaf: centre,
    // radius^2, sphere index, moving flag.  [6] inner: left ref16 | right
    // ref16 << 16.  [7] inner: 8 * axis.  [8..11] eight 16-bit exit links.  [12..14] (moving scenes) centre1 - centre0.
    std::vector<uint32_t> small_nodes;
    uint32_t small_words = 0;
    uint32_t small_root = kRef16End;

    std::vector<Float4> sph;        // {centre at time0, radius}
    std::vector<Float4> motion;     // {centre1 - centre0, time0}
    std::vector<float> motion_dt;   // time1 - time0
    std::vector<uint32_t> sph_info; // see kInfo*
    std::vector<Float4> mat;        // {albedo rgb, fuzz or ref_idx}
    std::vector<Float4> pmat;       // mat[material of sphere i]
    std::vector<Float4> prim_att;   // attenuation of a scatter off this sphere (materials.rs:32,55,101), and its fuzz or ref_idx
    uint32_t n_spheres = 0;
    uint32_t flags = 0;
    float u_time0 = 0.f, u_dt = 1.f;
};

// Returns RRT_OK or a negative rrt_status with a message in err.
int build_scene_layout(const rrt_sphere_soa *spheres, const rrt_material_soa *materials, const rrt_bvh_node *nodes,
                       uint32_t n_nodes, SceneLayout &out, std::string &err);

}  // namespace rrt_amd
