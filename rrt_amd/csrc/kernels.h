// Device-side data layout and launch entry points of the wavefront path tracer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rrt_amd {

constexpr uint32_t kWaveSlots = 256;     // path slots owned by one wave (4 tiles of 64 lanes)
constexpr uint32_t kBlockThreads = 256;  // 4 waves per workgroup
constexpr uint32_t kCursorShards = 64;   // sharded path cursors, one 128-byte line each
constexpr uint32_t kCursorStride = 32;   // uint32 words between two cursors
constexpr uint32_t kWaveStatWords = 16;  // uint64 counters per wave
constexpr uint32_t kPrimNone = 0xFFFFFFFFu;

// Indices into a wave's counter row
enum { ST_RAYS = 0, ST_INNER, ST_LEAF, ST_HL, ST_HM, ST_HD, ST_SKY, ST_PATHS, ST_DEPTHCUT, ST_ABSORBED };

struct CameraDev {  // camera.rs:5-16
    float llc[3], hor[3], ver[3], org[3], u[3], v[3];
    float lens_radius, time_low, time_scale;
};

struct SceneDev {
    const uint4 *nodes;  // 4 x uint4 per node (see scene_layout.h)
    uint32_t root_ref;
    const float4 *sph;        // {centre0, radius}
    const float4 *motion;     // {centre1 - centre0, time0}
    const float *motion_dt;   // time1 - time0
    const uint32_t *sph_info;
    const float4 *prim_att;   // per sphere: attenuation rgb, fuzz; glass: 1/ior, r0 front, r0 back, ior
    uint32_t flags;
    float u_time0, u_dt;
    // compact copy for LDS residency (null when the scene does not qualify; see scene_layout.h)
    const uint4 *small_nodes;
    uint32_t small_bytes;       // size of the image (the device copy is padded to a multiple of 16 bytes)
    uint32_t small_root;        // reference (byte offset) of the root
    uint32_t small_end;         // the reference that ends a walk; inner nodes lie below it, leaves above
    uint32_t small_leaf_links;  // byte offset of the eight 16-bit exit links inside a leaf
    uint32_t n_nodes;
    uint32_t n_spheres;
    // copy of the top levels of the tree for scenes without a compact copy (references with kRefTopBit index it), or null
    const uint4 *top_nodes;
    uint32_t n_top;
    // records of the two-children walk (scene_layout.h: SceneLayout::wide: the top-of-tree copy's first, inner nodes named
    // by record index) and the root named the same way; null when the scene has none (it fits in LDS, or its tree is deeper
    // than the walk's stack)
    const uint4 *wide;
    uint32_t wide_root;
    float root_box[6];  // the root's box as (min, max) per axis
};

// All per-path state lives in planes indexed by slot.  A wave owns slots [wave*kWaveSlots, +kWaveSlots)
// and the matching stretch of `queue`, which lists its live slots; the low half of `qcount[wave]` is the list length, the
// high half the number of entries at the front of the list whose paths have just started (their rays are camera rays).
// PathState::layout bits.  kLayoutTimed: the origin plane carries the ray's time in a fourth word (scenes with moving
// spheres; elsewhere nothing reads the time and the plane is packed, 12 bytes per slot).  kLayoutNarrow: depth, trail
// length and the four newest trail entries share one 8-byte record per slot in the `recent` plane (8 bits each for the
// counts, 10 bits per entry: scenes of at most 1024 spheres rendered with a depth limit of at most 255); otherwise a
// 4-byte `depth` plane and 16 bytes per slot in `recent`.
// kLayoutFreshApart: the shade kernel writes every list with the freshly started paths first (a full list too, which is
// otherwise left as it is), and an extend workgroup whose lists are all full takes its rays in the order "paths under way
// of all its lists, then the fresh ones".  Camera rays of neighbouring pixels walk the tree alike and end together; rays
// that have bounced do not, whatever they are sorted by.  With the camera rays last, the end of a workgroup's pool, where
// its waves run out of rays to take one lane after the other, is short.
constexpr uint32_t kLayoutTimed = 1u, kLayoutNarrow = 2u, kLayoutFreshApart = 4u;
constexpr uint32_t kListCountMask = 0xFFFFu, kListFreshShift = 16;
constexpr uint32_t kNarrowMaxDepth = 255;
constexpr uint32_t kShadeTableSpheres = 512;  // scenes up to this size: the shade kernel keeps the sphere tables in LDS and the path state narrow

struct PathState {
    float4 *rayA;      // origin xyz, time (kLayoutTimed) or origin xyz packed
    float4 *rayB;      // direction xyz, path id within the batch (as bits)
    uint4 *rng;        // Xoshiro128Plus state
    uint32_t *depth;   // scatters so far (low 16 bits) | trail entries written (high 16 bits)
    uint2 *hit;        // extend -> shade: {t bits, sphere index | material kind << 30, or kPrimNone}
    uint4 *recent;     // the four newest spheres the path scattered off (dielectrics excepted), newest first
    uint32_t *trail;   // [slot / 64][level][slot % 64] the older ones; both feed the back-to-front product of lib.rs:62
    uint32_t *queue;
    uint32_t *qcount;
    uint32_t n_slots;  // multiple of kWaveSlots
    uint32_t trail_levels;  // levels per path in `trail` (the depth limit)
    uint32_t wave_begin, wave_end;  // the waves (slot stretches) one launch works on
    float t_min, t_max;             // the interval Bvh::hit searches: 0.001 and f32::MAX for a render (lib.rs:57)
    uint32_t layout;                // kLayout* bits
};

// Division of a 32-bit number by an invariant divisor with one multiply-high (Granlund & Montgomery): for every n,
// n / d == (t + ((n - t) >> sh1)) >> sh2 with t = high 32 bits of magic * n.
struct FastDiv { uint32_t magic, sh1, sh2; };
inline FastDiv make_fastdiv(uint32_t d) {  // d >= 1
    uint32_t L = 0;
    while (L < 32 && (uint64_t(1) << L) < d) L++;  // ceil(log2 d)
    FastDiv f;
    f.magic = (uint32_t)(((uint64_t(1) << 32) * ((uint64_t(1) << L) - d)) / d + 1);
    f.sh1 = L < 1 ? L : 1;
    f.sh2 = L < 1 ? 0 : L - 1;
    return f;
}

struct BatchDev {
    uint32_t nx, ny;
    uint32_t npix;             // pixels this call renders
    FastDiv div_nx, div_npix;
    float inv_nx, inv_ny;      // 1.0f / nx and 1.0f / ny, correctly rounded (IEEE division on the host)
    const uint32_t *pixel_map; // local pixel -> image pixel index (lib.rs:93-94 idx); null = identity
    uint32_t n_paths;          // npix * samples in this batch
    uint32_t k_first, k_step;  // sample index of local sample j is k_first + j * k_step
    uint64_t seed;
    float4 *rad;               // radiance (r, g, b, unused) of each path of the batch, by path id
    uint32_t max_depth;
    uint32_t *cursors;         // n_shards counters, kCursorStride words apart
    uint32_t n_shards;         // min(kCursorShards, waves): every shard must be served by at least one wave
    uint32_t *alive_flag;      // null, or 32 words 128 bytes apart: a wave that still owns live paths after shading sets one
};

// One launch of a kernel with a 256-byte private segment: makes the runtime size the stream's scratch memory now rather
// than at the first render kernel that reserves any (for the entry point that must allocate nothing while it runs)
void launch_prime_scratch(hipStream_t s, uint32_t *any_device_word);
void launch_raygen(hipStream_t s, const PathState &P, const BatchDev &B, const CameraDev &C, uint64_t *wave_stats);
void launch_extend(hipStream_t s, const SceneDev &S, const PathState &P, bool moving, bool count, uint32_t retire_min,
                   uint32_t leaf_min, uint64_t *wave_stats);
void launch_shade(hipStream_t s, const SceneDev &S, const PathState &P, const BatchDev &B, const CameraDev &C, bool moving,
                  bool count, uint64_t *wave_stats);
// The rest of a batch in one launch: every wave runs extend and shade on its own list until the list is empty (at most
// max_rounds turns; a wave that stops with paths left raises alive_flag).  finish_fits says whether the scene and the
// state layout allow it (BVH image and tables in LDS together); launch_finish returns nonzero if HIP refused the set-up.
bool finish_fits(const SceneDev &S, const PathState &P);
// The live lists of all of P's waves (P.wave_begin == 0) concatenated and cut into lists of list_len (< kWaveSlots) entries
// for the first waves, in a second list plane (merged_queue: as large as P.queue; merged_qcount: one word per wave;
// offsets: one word per wave and one more): what the finishing launch runs on once the lists have thinned out.
void launch_merge_lists(hipStream_t s, const PathState &P, uint32_t *merged_queue, uint32_t *merged_qcount, uint32_t *offsets, uint32_t list_len);
int launch_finish(hipStream_t s, const SceneDev &S, const PathState &P, const BatchDev &B, const CameraDev &C, bool moving,
                  bool count, uint32_t max_rounds, uint32_t retire_min, uint32_t leaf_min, uint32_t *alive_flag, uint64_t *wave_stats);
// accum[3][npix] += radiance of the batch's paths, sample by sample in order (lib.rs:96-103)
void launch_accumulate(hipStream_t s, float *accum, uint32_t npix, const float4 *rad, uint32_t n_samples,
                       bool first_batch);
// lib.rs:105-108 for local pixel planes -> interleaved RGB8 / float sums at image positions
void launch_finalize(hipStream_t s, const float *accum, uint32_t npix, const uint32_t *pixel_map, uint32_t ns, uint8_t *rgb8,
                     float *accum_out);
// rgb8 from interleaved float sums (lib.rs:105-108)
void launch_finalize_interleaved(hipStream_t s, const float *accum_rgb, uint32_t n_pixels, uint32_t ns, uint8_t *rgb8);
// Bvh::hit alone on caller rays: fills the first n slots of the path pool (ray planes, identity live lists, list lengths) /
// reads their hit records back.  The walk itself is launch_extend, the kernel a render runs.
void launch_intersect_setup(hipStream_t s, const PathState &P, uint32_t n, const float *org, const float *dir, const float *time);
void launch_intersect_gather(hipStream_t s, const PathState &P, uint32_t n, float *out_t, uint32_t *out_obj);
// image[map[l]] = packed[l] for the l-th pixel of a tile shard (3 components per pixel): puts a rank's share in place
void launch_scatter_rgb8(hipStream_t s, const uint8_t *packed, const uint32_t *map, uint32_t n_pixels, uint8_t *image);
void launch_scatter_f32(hipStream_t s, const float *packed, const uint32_t *map, uint32_t n_pixels, float *image);
// dst[i] += src[i]
void launch_add_f32(hipStream_t s, float *dst, const float *src, size_t n);
void launch_reduce_stats(hipStream_t s, const uint64_t *wave_stats, uint32_t n_waves, uint64_t *out16);

}  // namespace rrt_amd
