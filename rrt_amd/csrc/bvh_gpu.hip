// Bvh::build (src/acceleration.rs:131-217) on the GPU: the same tree, node for node and bit for bit, as the recursive
// host builder (rrt_amd/host/rrt.cpp: Bvh::build_r), built level by level.  That is the reference's tree wherever its
// result is determined; where it leaves the order of tied keys to the library (select_nth_unstable_by in the
// equal-count split), ties fall as a stable sort puts them.
//
// What makes a level-synchronous build possible without knowing the tree in advance:
//   * a subtree over n objects has exactly 2n-1 nodes and the reference pushes a node before its left subtree and the
//     left subtree before the right one (:158,198), so the children of node i over (n_left, n_right) objects are nodes
//     i + 1 and i + 2 * n_left;
//   * the Middle split (:174-179) partitions a contiguous range by `centroid[axis] < midpoint`.  The reference's
//     partition swaps from both ends and is not stable, but which objects land on each side, and with them every box
//     and every later split, does not depend on their order inside the side; here a stable partition by one prefix sum
//     over the whole object array serves every open range of a level at once;
//   * box and centroid bounds of a range are min/max reductions, exact in any order.
//   * the equal-count split (:186-194) is "sort the range by centroid[axis], cut it in the middle": one stable radix sort
//     of the whole object array by (start of the object's range, centroid[axis]) sorts every such range of a level at
//     once and leaves everything else where it is.
// A range whose Middle split leaves one side empty switches, with its whole subtree, to the equal-count split
// (:180-183); a build can also ask for equal counts from the root down (BvhPartitionMethod::EqualCount).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rrt_amd.h"

namespace rrt_amd {
namespace {

constexpr uint32_t kClosed = 0xFFFFFFFFu;
constexpr uint32_t kNoNode = 0xFFFFFFFFu;
constexpr float kFltMax = 3.402823466e+38f;

struct ObjBox { float mn[3], mx[3], c[3]; };

struct Seg {  // an open range of the current level
    uint32_t start, count, node;
    uint32_t equal;  // split by equal counts (asked for, or inherited from an ancestor whose Middle split failed)
};

struct SegWork {
    uint32_t box_mn[3], box_mx[3], cen_mn[3], cen_mx[3];  // order-preserving integer images of the floats
    uint32_t n_left, axis;
    float mid;
    uint32_t scan_at_start;
    uint32_t left_seg, right_seg;    // index in the next level's list, or kClosed
    uint32_t left_node, right_node;  // node index of a child
    uint32_t equal;                  // this range is cut by equal counts (its objects are taken from the sorted order)
};

__host__ __device__ inline uint32_t enc(float f) {
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float dec(uint32_t u) {
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __builtin_bit_cast(float, b);
}

// sphere_aabb (src/sphere.rs:38-55, literal: no |radius|) and Aabb::centroid (src/acceleration.rs:28-30)
__host__ __device__ inline ObjBox object_box(const float c0[3], const float c1[3], float r) {
    ObjBox o;
    for (int k = 0; k < 3; k++) {
        const float a0 = c0[k] - r, a1 = c0[k] + r, b0 = c1[k] - r, b1 = c1[k] + r;
        o.mn[k] = a0 < b0 ? a0 : b0;   // Aabb::union: min_by_component / max_by_component, first operand on a tie
        o.mx[k] = a1 > b1 ? a1 : b1;
        o.c[k] = 0.5f * o.mn[k] + 0.5f * o.mx[k];
    }
    return o;
}

__global__ void k_prepare(uint32_t n, const float *c0x, const float *c0y, const float *c0z, const float *c1x, const float *c1y,
                          const float *c1z, const float *radius, ObjBox *boxes, uint32_t *order, uint32_t *seg_id) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c0[3] = {c0x[i], c0y[i], c0z[i]}, c1[3] = {c1x[i], c1y[i], c1z[i]};
    boxes[i] = object_box(c0, c1, radius[i]);
    order[i] = i;
    seg_id[i] = 0u;
}

__global__ void k_init_work(uint32_t n_seg, SegWork *work) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    SegWork w{};
    for (int k = 0; k < 3; k++) {  // Aabb::default(): minimum = f32::MAX, maximum = f32::MIN (src/acceleration.rs:10-14)
        w.box_mn[k] = enc(kFltMax); w.box_mx[k] = enc(-kFltMax);
        w.cen_mn[k] = enc(kFltMax); w.cen_mx[k] = enc(-kFltMax);
    }
    w.left_seg = w.right_seg = kClosed;
    w.left_node = w.right_node = kNoNode;
    work[s] = w;
}

// min/max of a value over the lanes of the wave that are active here
__device__ inline uint32_t wave_min(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}
__device__ inline uint32_t wave_max(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}

__global__ void k_bounds(uint32_t n, const uint32_t *seg_id, const uint32_t *order, const ObjBox *boxes, SegWork *work) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = i < n ? seg_id[i] : kClosed;
    // a wave whose 64 objects all sit in one range (the upper levels) reduces first and issues one atomic per value
    const uint32_t s0 = (uint32_t)__shfl((int)s, 0, 64);
    const bool uniform = __ballot(s != s0) == 0ull && __builtin_popcountll(__ballot(1)) == 64;
    if (uniform) {
        if (s0 == kClosed) return;
        const ObjBox b = boxes[order[i]];
        for (int k = 0; k < 3; k++) {
            const uint32_t a = wave_min(enc(b.mn[k])), c = wave_max(enc(b.mx[k])), d = wave_min(enc(b.c[k])), e = wave_max(enc(b.c[k]));
            if ((threadIdx.x & 63u) == 0u) {
                atomicMin(&work[s0].box_mn[k], a); atomicMax(&work[s0].box_mx[k], c);
                atomicMin(&work[s0].cen_mn[k], d); atomicMax(&work[s0].cen_mx[k], e);
            }
        }
        return;
    }
    if (s == kClosed) return;
    const ObjBox b = boxes[order[i]];
    for (int k = 0; k < 3; k++) {
        atomicMin(&work[s].box_mn[k], enc(b.mn[k])); atomicMax(&work[s].box_mx[k], enc(b.mx[k]));
        atomicMin(&work[s].cen_mn[k], enc(b.c[k])); atomicMax(&work[s].cen_mx[k], enc(b.c[k]));
    }
}

// axis = centroid_bounds.max_extent() (:167-170, src/acceleration.rs:36-45), midpoint (:175), and the node's box
__global__ void k_decide(uint32_t n_seg, const Seg *segs, SegWork *work, rrt_bvh_node *nodes) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    SegWork &w = work[s];
    const float dx = dec(w.cen_mx[0]) - dec(w.cen_mn[0]), dy = dec(w.cen_mx[1]) - dec(w.cen_mn[1]), dz = dec(w.cen_mx[2]) - dec(w.cen_mn[2]);
    const uint32_t axis = (dx > dy && dx > dz) ? 0u : (dy > dz ? 1u : 2u);
    w.axis = axis;
    w.mid = (dec(w.cen_mn[axis]) + dec(w.cen_mx[axis])) / 2.0f;
    rrt_bvh_node &nd = nodes[segs[s].node];
    for (int k = 0; k < 3; k++) { nd.min[k] = dec(w.box_mn[k]); nd.max[k] = dec(w.box_mx[k]); }
}

__global__ void k_side(uint32_t n, const uint32_t *seg_id, const uint32_t *order, const Seg *segs, const ObjBox *boxes, SegWork *work, uint32_t *flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = seg_id[i];
    uint32_t left = 0;
    if (s != kClosed && !segs[s].equal) {
        left = boxes[order[i]].c[work[s].axis] < work[s].mid ? 1u : 0u;  // :176
        if (left) atomicAdd(&work[s].n_left, 1u);
    }
    flag[i] = left;
}

__global__ void k_split(uint32_t n_seg, const Seg *segs, SegWork *work, const uint32_t *scan, rrt_bvh_node *nodes, Seg *next, uint32_t *n_next,
                        uint32_t *n_equal) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    SegWork &w = work[s];
    const Seg g = segs[s];
    w.scan_at_start = scan[g.start];
    const bool equal = g.equal != 0u || w.n_left == 0u || w.n_left == g.count;  // :180-183: equal counts from here down
    w.equal = equal ? 1u : 0u;
    if (equal) { w.n_left = g.count / 2u; atomicAdd(n_equal, 1u); }  // :187
    const uint32_t nl = w.n_left, nr = g.count - nl;
    w.left_node = g.node + 1u;
    w.right_node = g.node + 2u * nl;
    nodes[g.node].left_or_obj = w.left_node;
    nodes[g.node].right_axis_leaf = w.right_node | (w.axis << RRT_NODE_AXIS_SHIFT);
    if (nl >= 2u) { w.left_seg = atomicAdd(n_next, 1u); next[w.left_seg] = Seg{g.start, nl, w.left_node, w.equal}; }
    if (nr >= 2u) { w.right_seg = atomicAdd(n_next, 1u); next[w.right_seg] = Seg{g.start + nl, nr, w.right_node, w.equal}; }
}

// Sort keys of a level with equal-count ranges: (start of the range, centroid along the range's axis) for their objects,
// (own position, 0) for everything else, so that one stable sort of the whole array sorts each such range in place.
__global__ void k_sort_keys(uint32_t n, const uint32_t *seg_id, const uint32_t *order, const Seg *segs, const SegWork *work, const ObjBox *boxes, uint64_t *keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = seg_id[i];
    uint64_t k = (uint64_t)i << 32;
    if (s != kClosed && work[s].equal) k = ((uint64_t)segs[s].start << 32) | enc(boxes[order[i]].c[work[s].axis]);
    keys[i] = k;
}

// the stable partition (or the cut of the sorted order) of every open range, and the leaves of the ranges that are down to
// one object
__global__ void k_scatter(uint32_t n, const uint32_t *seg_id, const uint32_t *order, const uint32_t *order_sorted, const Seg *segs, const SegWork *work,
                          const uint32_t *scan, const uint32_t *flag, uint32_t *order_next, uint32_t *seg_next, rrt_bvh_node *nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = seg_id[i];
    if (s == kClosed) { order_next[i] = order[i]; seg_next[i] = kClosed; return; }
    const SegWork &w = work[s];
    const Seg g = segs[s];
    uint32_t obj, pos;
    bool left;
    if (w.equal) {  // position i of the range, in sorted order (:188-193)
        obj = order_sorted[i];
        pos = i;
        left = i - g.start < w.n_left;
    } else {
        obj = order[i];
        const uint32_t lefts_before = scan[i] - w.scan_at_start;
        left = flag[i] != 0u;
        pos = left ? g.start + lefts_before : g.start + w.n_left + ((i - g.start) - lefts_before);
    }
    const uint32_t child_seg = left ? w.left_seg : w.right_seg;
    order_next[pos] = obj;
    seg_next[pos] = child_seg;
    const uint32_t child_count = left ? w.n_left : g.count - w.n_left;
    if (child_count == 1u) {  // :146-156
        rrt_bvh_node leaf{};
        leaf.left_or_obj = obj;
        leaf.right_axis_leaf = RRT_NODE_LEAF_BIT;
        nodes[left ? w.left_node : w.right_node] = leaf;
    }
}

struct Buf {
    void *p = nullptr;
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    ~Buf() { if (p) (void)hipFree(p); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

}  // namespace

#define BVH_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return RRT_ERR_HIP; } \
    } while (0)

int gpu_build_bvh(hipStream_t st, const rrt_sphere_soa *sp, uint32_t method, rrt_bvh_node *out_nodes, uint32_t capacity, uint32_t *n_nodes, std::string &err) {
    const uint32_t n = sp->n;
    const uint32_t total = 2u * n - 1u;
    if (capacity < total) { err = "rrt_build_bvh: out_nodes holds fewer than 2n-1 nodes"; return RRT_ERR_INVALID_ARGUMENT; }
    if (n == 1) {
        rrt_bvh_node leaf{};
        leaf.left_or_obj = 0; leaf.right_axis_leaf = RRT_NODE_LEAF_BIT;
        out_nodes[0] = leaf; *n_nodes = 1;
        return RRT_OK;
    }
    Buf planes[7], boxes, order[2], seg_id[2], flag, scan, segs[2], work, nodes, counters, cub_tmp, keys[2], order_sorted;
    const float *src[7] = {sp->c0x, sp->c0y, sp->c0z, sp->c1x, sp->c1y, sp->c1z, sp->radius};
    for (int k = 0; k < 7; k++) {
        BVH_TRY(planes[k].alloc((size_t)n * 4));
        BVH_TRY(hipMemcpyAsync(planes[k].p, src[k], (size_t)n * 4, hipMemcpyHostToDevice, st));
    }
    BVH_TRY(boxes.alloc((size_t)n * sizeof(ObjBox)));
    for (int k = 0; k < 2; k++) { BVH_TRY(order[k].alloc((size_t)n * 4)); BVH_TRY(seg_id[k].alloc((size_t)n * 4)); BVH_TRY(segs[k].alloc((size_t)n * sizeof(Seg))); }
    BVH_TRY(flag.alloc((size_t)n * 4)); BVH_TRY(scan.alloc((size_t)n * 4));
    BVH_TRY(work.alloc((size_t)n * sizeof(SegWork)));
    BVH_TRY(nodes.alloc((size_t)total * sizeof(rrt_bvh_node)));
    BVH_TRY(counters.alloc(8));
    size_t tmp_bytes = 0, sort_bytes = 0;
    BVH_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, flag.as<uint32_t>(), scan.as<uint32_t>(), (int)n, st));
    // (the sort's buffers are made when a level first needs them: a Middle build of distinct centroids never does)
    bool sort_ready = false;

    const dim3 blk(256), grid_n((n + 255u) / 256u);
    hipLaunchKernelGGL(k_prepare, grid_n, blk, 0, st, n, planes[0].as<float>(), planes[1].as<float>(), planes[2].as<float>(), planes[3].as<float>(),
                       planes[4].as<float>(), planes[5].as<float>(), planes[6].as<float>(), boxes.as<ObjBox>(), order[0].as<uint32_t>(),
                       seg_id[0].as<uint32_t>());
    const Seg root{0u, n, 0u, method == RRT_BVH_EQUAL_COUNT ? 1u : 0u};
    BVH_TRY(hipMemcpyAsync(segs[0].p, &root, sizeof(Seg), hipMemcpyHostToDevice, st));
    uint32_t *n_next = counters.as<uint32_t>(), *n_equal = counters.as<uint32_t>() + 1;
    BVH_TRY(cub_tmp.alloc(tmp_bytes));

    uint32_t n_seg = 1;
    int cur = 0;
    for (uint32_t level = 0; n_seg != 0u; level++) {
        if (level > n) { err = "rrt_build_bvh: level loop did not terminate"; return RRT_ERR_HIP; }
        const dim3 grid_s((n_seg + 255u) / 256u);
        BVH_TRY(hipMemsetAsync(counters.p, 0, 8, st));
        hipLaunchKernelGGL(k_init_work, grid_s, blk, 0, st, n_seg, work.as<SegWork>());
        hipLaunchKernelGGL(k_bounds, grid_n, blk, 0, st, n, seg_id[cur].as<uint32_t>(), order[cur].as<uint32_t>(), boxes.as<ObjBox>(), work.as<SegWork>());
        hipLaunchKernelGGL(k_decide, grid_s, blk, 0, st, n_seg, segs[cur].as<Seg>(), work.as<SegWork>(), nodes.as<rrt_bvh_node>());
        hipLaunchKernelGGL(k_side, grid_n, blk, 0, st, n, seg_id[cur].as<uint32_t>(), order[cur].as<uint32_t>(), segs[cur].as<Seg>(), boxes.as<ObjBox>(),
                           work.as<SegWork>(), flag.as<uint32_t>());
        BVH_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp.p, tmp_bytes, flag.as<uint32_t>(), scan.as<uint32_t>(), (int)n, st));
        hipLaunchKernelGGL(k_split, grid_s, blk, 0, st, n_seg, segs[cur].as<Seg>(), work.as<SegWork>(), scan.as<uint32_t>(), nodes.as<rrt_bvh_node>(),
                           segs[cur ^ 1].as<Seg>(), n_next, n_equal);
        BVH_TRY(hipGetLastError());
        uint32_t got[2] = {0, 0};  // ranges of the next level; ranges of this one cut by equal counts
        BVH_TRY(hipMemcpyAsync(got, counters.p, 8, hipMemcpyDeviceToHost, st));
        BVH_TRY(hipStreamSynchronize(st));
        if (got[1] != 0u) {
            if (!sort_ready) {
                BVH_TRY(keys[0].alloc((size_t)n * 8)); BVH_TRY(keys[1].alloc((size_t)n * 8)); BVH_TRY(order_sorted.alloc((size_t)n * 4));
                BVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys[0].as<uint64_t>(), keys[1].as<uint64_t>(), order[0].as<uint32_t>(),
                                                           order_sorted.as<uint32_t>(), (int)n, 0, 64, st));
                if (sort_bytes > tmp_bytes) { Buf bigger; BVH_TRY(bigger.alloc(sort_bytes)); std::swap(bigger.p, cub_tmp.p); tmp_bytes = sort_bytes; }
                sort_ready = true;
            }
            hipLaunchKernelGGL(k_sort_keys, grid_n, blk, 0, st, n, seg_id[cur].as<uint32_t>(), order[cur].as<uint32_t>(), segs[cur].as<Seg>(), work.as<SegWork>(),
                               boxes.as<ObjBox>(), keys[0].as<uint64_t>());
            size_t bytes = tmp_bytes;
            BVH_TRY(hipcub::DeviceRadixSort::SortPairs(cub_tmp.p, bytes, keys[0].as<uint64_t>(), keys[1].as<uint64_t>(), order[cur].as<uint32_t>(),
                                                       order_sorted.as<uint32_t>(), (int)n, 0, 64, st));
        }
        hipLaunchKernelGGL(k_scatter, grid_n, blk, 0, st, n, seg_id[cur].as<uint32_t>(), order[cur].as<uint32_t>(),
                           got[1] != 0u ? order_sorted.as<uint32_t>() : order[cur].as<uint32_t>(), segs[cur].as<Seg>(), work.as<SegWork>(), scan.as<uint32_t>(),
                           flag.as<uint32_t>(), order[cur ^ 1].as<uint32_t>(), seg_id[cur ^ 1].as<uint32_t>(), nodes.as<rrt_bvh_node>());
        BVH_TRY(hipGetLastError());
        n_seg = got[0];
        cur ^= 1;
    }
    BVH_TRY(hipMemcpyAsync(out_nodes, nodes.p, (size_t)total * sizeof(rrt_bvh_node), hipMemcpyDeviceToHost, st));
    BVH_TRY(hipStreamSynchronize(st));
    *n_nodes = total;
    return RRT_OK;
}

}  // namespace rrt_amd
