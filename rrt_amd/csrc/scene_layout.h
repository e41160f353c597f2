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
constexpr uint32_t kRefIndexMask = 0x0FFFFFFFu; // node index inside a reference
constexpr uint32_t kRefTopBit = 0x10000000u;    // the index counts inside the top-of-tree copy (make_top_tree), not the node array
constexpr uint32_t kRefMask = kRefLeafBit | kRefTopBit | kRefIndexMask;  // a reference without the axis bits that share word 6
constexpr uint32_t kTopMaxNodes = 1024;         // 64 KB of LDS
constexpr uint32_t kNodeWords = 16;             // 64 bytes
constexpr uint32_t kLeafPrimMask = 0x3FFFFFFFu; // leaf word 4: sphere index | material kind << 30
constexpr uint32_t kLeafKindShift = 30;

// Compact layout for scenes small enough to live in a workgroup's LDS.  A 16-bit reference is the node's byte offset
// in the image (= its LDS address: the image starts at LDS address 0).  Inner nodes (80 bytes) come first, then one
// unused word whose offset is the reference that means "nothing left to visit", then the leaves (36 bytes, or 52 in a
// scene with moving spheres); so `ref < end` is an inner node and `ref > end` a leaf.  The links INSIDE the image (near
// child, exit links) are made for the octant that follows them: one that leads to an inner node is that node's offset plus
// 8 for the octants whose rays are negative along x, i.e. the address of the (near, far) pair of its x planes for the ray,
// so that the walk reads the pair at the reference itself and reaches the other two pairs and the link word by three
// per-ray offsets from there, not four from the node's start.  small_root is the plain offset.
constexpr uint32_t kSmallInnerBytes = 80;
constexpr uint32_t kSmallLeafBytes = 36;        // centre, radius^2, sphere word, 8 exit links of 16 bits
constexpr uint32_t kSmallLeafBytesMoving = 52;  // + moving flag and centre1 - centre0
constexpr uint32_t kSmallLinkOffset = 48;       // inner: byte offset of the 8 (near child | exit << 16) words
constexpr uint32_t kSmallLdsBudget = 0xFFF0u;   // every offset fits 16 bits

// Per-sphere info word: material index | kind << 28 | moving << 31
constexpr uint32_t kInfoMatMask = 0x0FFFFFFFu;
constexpr uint32_t kInfoKindShift = 28;
constexpr uint32_t kInfoMovingBit = 0x80000000u;

// Scene flags
constexpr uint32_t kSceneHasMoving = 1u;
constexpr uint32_t kSceneUniformTimes = 2u;
// every centre component (both ends of a moving sphere's way) is at most 2^29 in magnitude and every radius at most 2^30,
// and spheres that move share one shutter interval
constexpr uint32_t kSceneBounded = 4u;
// every radius has a magnitude in [2^-30, 2^30]: the shade kernel may divide by it through its reciprocal
constexpr uint32_t kSceneRadiiNormal = 8u;

// (set by make_wide_nodes) the layout has the records of the two-children walk: the nodes stay in HBM and the tree is no
// deeper than that walk's stack
constexpr uint32_t kSceneWideRecords = 16u;

struct Float4 { float x, y, z, w; };

struct SceneLayout {
    // Node words.  Inner: [0..5] box as (min, max) per axis: x, y, z, [6] right ref | axis << 29, [7] left ref,
    // [8..15] exit link per octant.  Leaf: [0..2] centre at time0, [3] radius^2, [4] sphere index | material kind << 30,
    // [5] bit 0 = moving, [8..15] exit link per octant.
    std::vector<uint32_t> nodes;
    std::vector<uint8_t> node_kind;  // per node: 0 = not reachable from the root, 1 = inner, 2 = leaf
    uint32_t n_nodes = 0;
    uint32_t root_ref = kRefEnd;
    uint32_t max_leaf_depth = 0;

    // Compact image (empty when the scene does not qualify), as words.
    // Inner: [0..11] per axis (min, max, max, min), so that the two words starting at 4 * axis + 2 * s are (near, far) for a
    //        ray whose direction is negative (s = 1) or not (s = 0) along that axis, and either pair is 8-byte aligned: one
    //        ds_read_b64 (2 LDS cycles per wave) instead of the ds_read2_b32 (4 cycles) a pair at a 4-byte boundary takes;
    //        [12..19] for each of the 8 sign octants of the ray direction: reference of the child the reference visits
    //        first (acceleration.rs:253-267) in the low half, exit link in the high half.
    // Leaf:  [0..2] centre at time0, [3] radius^2, [4] sphere index | kind << 30, then (scenes with moving spheres)
    //        [5] bit 0 = moving, [6..8] centre1 - centre0, then four words of eight 16-bit exit links.
    std::vector<uint32_t> small_nodes;
    uint32_t small_bytes = 0;       // size of the image
    uint32_t small_root = 0;        // reference of the root
    uint32_t small_end = 0;         // the reference that ends a walk
    uint32_t small_leaf_links = 0;  // byte offset of the exit links inside a leaf

    // Top of the tree (whole levels from the root, at most kTopMaxNodes nodes) in the 64-byte node format, for scenes
    // whose nodes stay in HBM: the extend kernel keeps this copy in LDS.  Filled by make_top_tree.
    std::vector<uint32_t> top_nodes;

    // Records of the two-children walk of big scenes (filled by make_wide_nodes, after make_top_tree; empty if the tree is
    // deeper than the walk's stack): one 64-byte record per inner node, in one array: first the records of the nodes of the
    // top-of-tree copy, in its order, then one per node of `nodes`.  Words [0..5] describe the left child and [6..11] the
    // right child: an inner child's box as (min, max) per axis, a leaf child's sphere as in a leaf node (centre at time0,
    // radius^2, sphere word, moving flag); [12] left child, [13] right child | axis << 29, where an inner child is named by
    // the index of its record in this array and a leaf child by its node reference.  `wide_root`: the root, named the same
    // way.  `root_box`: the root's own box (no record holds it).
    std::vector<uint32_t> wide;
    uint32_t wide_root = kRefEnd;
    float root_box[6] = {0, 0, 0, 0, 0, 0};

    std::vector<Float4> sph;        // {centre at time0, radius}
    std::vector<Float4> motion;     // {centre1 - centre0, time0}
    std::vector<float> motion_dt;   // time1 - time0
    std::vector<uint32_t> sph_info; // see kInfo*
    std::vector<Float4> mat;        // {albedo rgb, fuzz or ref_idx}
    std::vector<Float4> prim_att;   // attenuation of a scatter off this sphere (materials.rs:32,55) and its fuzz; glass: {1/ior, r0 front, r0 back, ior}
    uint32_t n_spheres = 0;
    uint32_t flags = 0;
    float u_time0 = 0.f, u_dt = 1.f;
};

// For a layout without a compact copy: picks the top levels of the tree, copies their nodes to `top_nodes` and rewrites
// every reference to one of them, in `nodes`, `top_nodes` and `root_ref`, as kRefTopBit | index in `top_nodes`.
void make_top_tree(SceneLayout &layout);

// Builds `wide`, `wide_root` and `root_box` from `nodes` / `top_nodes` (call after make_top_tree).
constexpr uint32_t kWideStackDepth = 64;  // deferred children a ray can hold: one per level of the tree at most
void make_wide_nodes(SceneLayout &layout);

// Returns RRT_OK or a negative rrt_status with a message in err.
int build_scene_layout(const rrt_sphere_soa *spheres, const rrt_material_soa *materials, const rrt_bvh_node *nodes,
                       uint32_t n_nodes, SceneLayout &out, std::string &err);

}  // namespace rrt_amd
