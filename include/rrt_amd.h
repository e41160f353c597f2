/*
 * rrt_amd.h — C ABI of the MI355X (gfx950) render path for jfroy/rrt.
 *
 * The reference has exactly one entry into its hot path:
 *
 *     pub fn tracescene(nx, ny, ns, scene: &Scene, camera: &Camera,
 *                       pool: &rayon::ThreadPool, pdc: &AtomicUsize) -> Vec<u8>      (src/lib.rs:73-81)
 *
 * and no FFI of its own.  This header is what a thin `-sys` crate would bind so that tracescene()
 * keeps its signature while the per-pixel ray generation, BVH traversal, sphere intersection and
 * material shading run on the GPU.  The host flattens `Scene.spheres` (src/scene.rs:5-8,
 * src/sphere.rs:9-17) into planes, `Bvh.nodes` (src/acceleration.rs:96-116) into a linear array in
 * the order Bvh::build_r pushed them (src/acceleration.rs:150-217), and `Camera`
 * (src/camera.rs:5-16) into rrt_camera, then calls the functions below.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returning int returns
 * RRT_OK (0) or a negative rrt_status; the message for the calling thread's last failure is
 * available from rrt_last_error().  A context belongs to one GPU and is thread-compatible, not
 * thread-safe.  All calls block until their results are complete, like the function they replace.
 * There is no CPU fallback: without a usable HIP device rrt_create fails.
 */
#ifndef RRT_AMD_H
#define RRT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_API __attribute__((visibility("default")))

typedef enum rrt_status {
    RRT_OK = 0,
    RRT_ERR_INVALID_ARGUMENT = -1,
    RRT_ERR_HIP = -2,           /* a HIP runtime call failed; message carries hipGetErrorString */
    RRT_ERR_NO_DEVICE = -3,
    RRT_ERR_BAD_SCENE = -4,     /* malformed BVH / indices out of range / empty scene */
    RRT_ERR_OUT_OF_MEMORY = -5,
    RRT_ERR_NOT_READY = -6      /* render before scene or camera were set */
} rrt_status;

/* Material kinds: the three `impl Material` of src/materials.rs:19,44,75. */
enum { RRT_MAT_LAMBERTIAN = 0, RRT_MAT_METAL = 1, RRT_MAT_DIELECTRIC = 2 };

/* Sphere planes; replaces Vec<Sphere> (src/sphere.rs:9-17).  A StationarySphere
 * (src/sphere.rs:57-59) is c1 == c0, time0 = 0, time1 = 1.  radius may be negative
 * (src/chap11.rs:38-42); it is used exactly as the reference uses it. */
typedef struct rrt_sphere_soa {
    const float *c0x, *c0y, *c0z;
    const float *c1x, *c1y, *c1z;
    const float *time0, *time1;
    const float *radius;
    const uint32_t *material_id; /* index into rrt_material_soa */
    uint32_t n;
} rrt_sphere_soa;

/* Material planes; replaces Box<dyn Material + Sync> (src/sphere.rs:15).  fuzz is read for METAL
 * (src/materials.rs:39-42), ref_idx for DIELECTRIC (src/materials.rs:71-73), albedo for
 * LAMBERTIAN and METAL. */
typedef struct rrt_material_soa {
    const uint32_t *kind;
    const float *albedo_r, *albedo_g, *albedo_b;
    const float *fuzz;
    const float *ref_idx;
    uint32_t n;
} rrt_material_soa;

/* One BvhNode (src/acceleration.rs:106-116), 32 bytes, in the reference's push order
 * (node 0 is the root; an Inner node's left child is the node that follows it).
 *   Inner: min/max = aabb, left_or_obj = left_i, right_axis_leaf = right_i | axis << 29.
 *   Leaf : left_or_obj = obj_i, right_axis_leaf = 1u << 31; min/max ignored. */
typedef struct rrt_bvh_node {
    float min[3];
    float max[3];
    uint32_t left_or_obj;
    uint32_t right_axis_leaf;
} rrt_bvh_node;

#define RRT_NODE_LEAF_BIT 0x80000000u
#define RRT_NODE_AXIS_SHIFT 29
#define RRT_NODE_INDEX_MASK 0x1FFFFFFFu  /* field width of the interchange node; rrt_upload_scene takes at most 2^28-1 nodes */

/* Camera (src/camera.rs:5-16) after Camera::from (src/camera.rs:31-54).  time_low / time_scale are
 * the two fields of rand's Uniform<f32> built by new_inclusive(time0, time1) (src/camera.rs:52). */
typedef struct rrt_camera {
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float origin[3];
    float u[3];
    float v[3];
    float lens_radius;
    float time_low;
    float time_scale;
} rrt_camera;

/* How one render call is cut out of the whole image, for multi-GPU runs.  With world = 1 the call
 * renders everything.  TILE: the image is cut into 32x8-pixel tiles, numbered row-major; this
 * call renders tiles t with t % world == rank and writes only their pixels.  SAMPLE: this call
 * renders samples k with k % world == rank of every pixel; accumulators of all ranks must be
 * summed before finalising (rrt_finalize). */
enum { RRT_SHARD_NONE = 0, RRT_SHARD_TILE = 1, RRT_SHARD_SAMPLE = 2 };

typedef struct rrt_render_params {
    uint32_t nx, ny, ns;  /* lib.rs:74-76 */
    uint32_t max_depth;   /* the literal 50 of lib.rs:58.  Taken as given: 0 is a depth limit of 0 (every hit returns black,
                             the image is sky and black), not "library default" — set it explicitly */
    uint64_t seed;        /* base of the per-(pixel, sample) Xoshiro128Plus streams */
    uint32_t shard_mode, shard_rank, shard_world;
    uint32_t count_work;  /* non-zero: also count node visits / sphere tests / shaded hits (slower build of the same kernels) */
    uint32_t path_slots;  /* paths in flight; 0 = library default */
    uint32_t batch_samples; /* samples per accumulation batch; 0 = library default */
    /* Measurement and tuning knobs, all 0 by default: [0] = n: time every n-th extend launch with HIP events (n in 1, 2,
     * 4, 8; fills rrt_stats.extend_ms / extend_launches); [1] bit 0: walk the HBM copy of the BVH even if the scene fits
     * LDS, bit 1: walk big scenes node by node along the exit links instead of two children per record; [2], [3]: lanes that must be idle / wait at a sphere before the extend kernel hands out rays / runs a sphere
     * pass (0 = library default).  None of them changes a bit of the result. */
    uint32_t reserved[4];
} rrt_render_params;

/* Work counters of one render call.  rays = calls of Bvh::hit (lib.rs:57). */
typedef struct rrt_stats {
    uint64_t rays;
    uint64_t inner_visits; /* Aabb::intersect calls (acceleration.rs:250); count_work only */
    uint64_t leaf_tests;   /* Sphere::hit calls (acceleration.rs:239);      count_work only */
    uint64_t hits_lambertian, hits_metal, hits_dielectric; /* scatter calls by material; count_work only */
    uint64_t sky_misses;   /* count_work only */
    uint64_t paths;
    uint64_t depth_cut;    /* lib.rs:58-60; count_work only */
    uint64_t absorbed;     /* lib.rs:64;    count_work only */
    uint64_t iterations;   /* extend+shade rounds launched */
    double   gpu_ms;       /* device time of the whole call between HIP events */
    double   extend_ms;    /* summed device time of the extend kernel launches (count_work or timing enabled) */
    uint64_t extend_launches;
    uint64_t debug[6];     /* count_work only: extend descent passes, lanes in them, sphere-test passes, lanes in them, retire passes, lanes */
} rrt_stats;

typedef struct rrt_ctx rrt_ctx;

RRT_API const char *rrt_last_error(void);
RRT_API const char *rrt_version(void);

/* One context per GPU.  device < 0 selects the current HIP device. */
RRT_API int rrt_create(int device, rrt_ctx **out);
RRT_API void rrt_destroy(rrt_ctx *ctx);

/* Replaces the `&Scene` argument of tracescene (lib.rs:77): validates the tree, builds the
 * device traversal layout and copies it to HBM.  The arrays may be freed after the call. */
RRT_API int rrt_upload_scene(rrt_ctx *ctx, const rrt_sphere_soa *spheres, const rrt_material_soa *materials,
                             const rrt_bvh_node *nodes, uint32_t n_nodes);

/* Replaces the `&Camera` argument of tracescene (lib.rs:78). */
RRT_API int rrt_set_camera(rrt_ctx *ctx, const rrt_camera *camera);

/* Replaces the body of tracescene (lib.rs:82-112).  out_rgb8: nx*ny*3 bytes, RGB interleaved, row 0
 * = top of the image (lib.rs:93-94,106-108), host memory.  out_accum (nullable): nx*ny*3 floats,
 * the per-pixel sums before the division by ns (lib.rs:95-103).  progress (nullable) plays the part
 * of `pdc` (lib.rs:81,109; polled by main.rs:78-94): the call zeroes it, raises it while it runs —
 * after every few extend/shade rounds, to the number of pixels' worth of finished samples — and
 * leaves it at the number of pixels this call rendered (nx*ny unless TILE-sharded); another thread
 * may read it meanwhile.  Under TILE sharding only this rank's pixels are written (the rest of both
 * buffers is zero-filled); under SAMPLE sharding out_accum holds this rank's partial sums and
 * out_rgb8 is the image of this rank's samples alone (quantised with this rank's sample count). */
RRT_API int rrt_render(rrt_ctx *ctx, const rrt_render_params *params, uint8_t *out_rgb8, float *out_accum,
                       rrt_stats *stats, volatile uint64_t *progress);

/* The same call on several GPUs of one node: ctxs[r] is rank r's context (one per GPU, each with the scene uploaded and the
 * camera set), the whole image comes back in host memory as from rrt_render (lib.rs:73-81: one call, whole image).
 * params->shard_mode picks how the ranks share the work: RRT_SHARD_TILE (or NONE: the default) — 32x8 tiles dealt
 * round-robin, every rank sends the pixels it owns to rank 0's GPU (ncclSend/ncclRecv in one group; bit-identical to a
 * one-GPU render); RRT_SHARD_SAMPLE — sample k goes to rank k mod n, the f32 sums are ncclReduce'd onto rank 0 and
 * quantised there (another order of f32 additions: within rounding of the one-GPU sums).  shard_rank / shard_world are
 * ignored.  stats are summed over the ranks (gpu_ms: the slowest rank).  One host thread per context for the call's
 * duration; RCCL is loaded on first use, and not at all with n_ctx == 1.  Environment, for tests on a one-GPU machine:
 * RRT_MULTI_LOOPBACK=1 lets contexts share a GPU and moves the shares with plain copies instead of RCCL;
 * RRT_MULTI_FORCE_RCCL=1 sends even a single rank's share through RCCL. */
RRT_API int rrt_render_multi(rrt_ctx *const *ctxs, uint32_t n_ctx, const rrt_render_params *params, uint8_t *out_rgb8,
                             float *out_accum, rrt_stats *stats, volatile uint64_t *progress);

/* Same, results left in device memory owned by the caller (hipMalloc'd / torch tensors):
 * d_rgb8 nx*ny*3 bytes (nullable), d_accum nx*ny*3 floats (nullable).  Runs on the context's
 * stream and returns after the stream has drained. */
RRT_API int rrt_render_device(rrt_ctx *ctx, const rrt_render_params *params, void *d_rgb8, void *d_accum,
                              rrt_stats *stats);

/* rrt_render_device for callers that cannot allocate device memory: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  The path pool,
 * queues, counters, radiance planes and per-pixel sums live in a static array of
 * RRT_NOALLOC_WORKSPACE_BYTES that is part of the library's code object (every GPU the library runs on
 * holds it until the process ends); the call makes no hipMalloc / hipFree and creates no events.
 * Limits, rejected with RRT_ERR_INVALID_ARGUMENT: nx*ny <= 512*512, max_depth <= 64, path_slots <= 16384
 * (0 = 16384), shard_mode NONE or SAMPLE, no kernel timing (reserved[0] is ignored).  Same image, bit for
 * bit, as rrt_render_device (scenes too big for LDS are walked along exit links here: the default walk of such scenes
 * keeps a stack in scratch memory, which the runtime allocates when its kernel is first launched; rrt_create runs one small
 * kernel with a private segment larger than any kernel's that this entry point launches on each of the context's streams,
 * so that the runtime sizes their scratch memory there and not inside a call of this entry point).  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two contexts, two threads) race; calls one after another do not,
 * each clears what it needs. */
#define RRT_NOALLOC_WORKSPACE_BYTES (16u << 20)
RRT_API int rrt_render_device_noalloc(rrt_ctx *ctx, const rrt_render_params *params, void *d_rgb8, void *d_accum,
                                      rrt_stats *stats);

/* lib.rs:105-108 on device buffers: rgb8 = saturate_u8(sqrt(accum / ns) * 255.99).  Used after
 * accumulators of SAMPLE shards have been summed across GPUs. */
RRT_API int rrt_finalize(rrt_ctx *ctx, const void *d_accum, uint32_t n_pixels, uint32_t ns, void *d_rgb8);

/* Closest hits of caller-supplied rays through the uploaded scene: Bvh::hit (acceleration.rs:221-272)
 * alone.  Host arrays; org/dir are n x 3, time is n (nullable = 0).  out_obj is 0xFFFFFFFF on a miss. */
RRT_API int rrt_intersect(rrt_ctx *ctx, uint32_t n, const float *org, const float *dir, const float *time,
                          float t_min, float t_max, float *out_t, uint32_t *out_obj);

/* Host-only checks of rrt_upload_scene without a GPU: same status codes and messages. */
RRT_API int rrt_validate_scene(const rrt_sphere_soa *spheres, const rrt_material_soa *materials, const rrt_bvh_node *nodes,
                               uint32_t n_nodes);

/* Host-only: the traversal layout rrt_upload_scene would copy to the GPU.  words16: n_nodes x 16 uint32
 * (inner: box as (min, max) pairs for x, y, z, right ref | axis << 29, left ref, 8 exit links; leaf: centre xyz, radius^2,
 * sphere index, moving flag, 0, 0, 8 exit links).  A ref is a node index, bit 31 set for a leaf,
 * 0x7FFFFFFF for "traversal finished".  scene_flags: bit 0 moving spheres, bit 1 one shutter interval for all of them,
 * bit 2 centres and radii within the range of the short division, bit 3 radii in [2^-30, 2^30], bit 4 the scene gets the
 * records of the two-children walk (nodes stay in HBM, tree no deeper than 64).  For inspection and for tests. */
RRT_API int rrt_debug_layout(const rrt_sphere_soa *spheres, const rrt_material_soa *materials, const rrt_bvh_node *nodes,
                             uint32_t n_nodes, uint32_t *words16, uint32_t *root_ref, uint32_t *scene_flags);

/* Host-only: image pixel indices (lib.rs:93-94 idx) a TILE shard renders, in the order the library
 * enumerates them; out_idx may be null to query the count.  With mode NONE / SAMPLE: all nx*ny pixels. */
RRT_API uint32_t rrt_shard_pixels(uint32_t nx, uint32_t ny, uint32_t shard_mode, uint32_t rank, uint32_t world, uint32_t *out_idx);

/* Bvh::build with BvhPartitionMethod::Middle (src/acceleration.rs:131-217; scene.rs:33-35 build_bvh) on the context's
 * GPU: 2n-1 nodes in the reference's push order over objects 0..n-1 (what rrt_upload_scene takes as `nodes`), identical
 * to the recursive builder's wherever its result is determined.  Where it is not — inside subtrees that fall back to
 * the equal-count split, select_nth_unstable_by leaves the order of equal keys to the implementation — ties are
 * resolved as a stable sort would.  Sphere planes and out_nodes are host memory; `capacity` must be at least 2n-1;
 * n <= 2^27 (the node index space of rrt_upload_scene).  Blocking. */
RRT_API int rrt_build_bvh(rrt_ctx *ctx, const rrt_sphere_soa *spheres, rrt_bvh_node *out_nodes, uint32_t capacity, uint32_t *n_nodes);

/* The same with the method of the top-level call (enum BvhPartitionMethod, src/acceleration.rs:98-103): RRT_BVH_MIDDLE is
 * rrt_build_bvh; RRT_BVH_EQUAL_COUNT sorts every range by centroid[axis] and cuts it at len / 2 (src/acceleration.rs:186-194),
 * ties in sorted order as a stable sort leaves them.  Either way the whole tree is built on the GPU, the equal-count
 * subtrees of a Middle build included. */
enum { RRT_BVH_MIDDLE = 0, RRT_BVH_EQUAL_COUNT = 1 };
RRT_API int rrt_build_bvh_method(rrt_ctx *ctx, const rrt_sphere_soa *spheres, uint32_t method, rrt_bvh_node *out_nodes, uint32_t capacity,
                                 uint32_t *n_nodes);

/* Name, compute units and HBM bytes of the context's device. */
RRT_API int rrt_device_info(rrt_ctx *ctx, char *name, size_t name_len, int *compute_units, uint64_t *hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* RRT_AMD_H */
