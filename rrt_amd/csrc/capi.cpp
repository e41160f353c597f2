// C ABI of include/rrt_amd.h: context, scene upload and the wavefront render loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <string>
#include <vector>

#include "../../include/rrt_amd.h"
#include "kernels.h"
#include "scene_layout.h"

using namespace rrt_amd;

namespace rrt_amd {
int gpu_build_bvh(hipStream_t st, const rrt_sphere_soa *sp, uint32_t method, rrt_bvh_node *out_nodes, uint32_t capacity, uint32_t *n_nodes, std::string &err);
// noalloc.hip: address and size of the current device's copy of the library's static workspace
int static_workspace(int device, void **ptr, size_t *bytes, std::string &err);
// for noalloc.hip: a render call that takes all its device memory from `workspace`
int render_in_workspace(rrt_ctx *ctx, const rrt_render_params *params, void *d_rgb8, void *d_accum, rrt_stats *stats, void *workspace, size_t workspace_bytes);
int context_device(const rrt_ctx *ctx);
int set_error(int code, const std::string &msg);
}

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            return fail(e_ == hipErrorOutOfMemory ? RRT_ERR_OUT_OF_MEMORY : RRT_ERR_HIP,                    \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                 \
        }                                                                                                   \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

constexpr uint32_t kDefaultSlots = 14336u * kWaveSlots;  // 3.5 Mi: the headline render is 1-2 % faster than with 4 Mi, the other configurations within 1 % (profiles/r03_sweep_slots.txt)
// per-batch radiance planes; HBM is 288 GB (capped by half of what is free).  Larger batches were measured and buy nothing:
// 3840x2160x1024 takes 1.38 s in two batches of 64 GiB as in eight of 16 GiB.
constexpr size_t kRadBudgetBytes = size_t(16) << 30;
constexpr uint32_t kMaxBatchPaths = 1u << 30;
constexpr uint32_t kMaxParts = 4;       // parts of the path pool that run on streams of their own
// measured: on scenes whose BVH is walked in LDS two parts beat one by 13 % and four by 2 %; on the 1 M-sphere scene
// (nodes in HBM, latency-bound extend) four beat two by 5 %.  With moving spheres the LDS image is 64 KB and no shade
// workgroup fits on a CU beside two extend workgroups: three parts beat two by 2.4 % there (book2chap2)
constexpr uint32_t kDefaultPartsLds = 2, kDefaultPartsLdsMoving = 3, kDefaultPartsHbm = 4;
// Before the finishing launch (render_impl): plain rounds after the last path is dealt; entries per merged list (below
// kWaveSlots: a list of kWaveSlots entries is a wave's own slots in order); pools too small for merging to matter
constexpr uint32_t kFinishDelayRounds = 3, kMergedListLength = kWaveSlots - 1, kMergeMinWaves = 1024;
constexpr uint32_t kFlagWords = kMaxParts * 32 * 32;  // per part: 32 flag words, one per 128-byte line
// extend+shade rounds queued between two looks at the alive flags: the first chunk of a render is short (a small render
// is over after a few rounds), every next one twice as long up to kMaxChunk: each look drains the device (-2.5 % step time
// at 32 against 8 on the headline render, whose 384 rounds then take 14 looks instead of 48; another -0.5 % at 64, and
// nothing more beyond: 128 and 256 measure within 0.1 ms of 64)
constexpr uint32_t kChunk = 8, kMaxChunk = 64;
constexpr uint32_t kMaxDepthLimit = 4096;  // depth and trail length share one 32-bit word per path
// Limits of rrt_render_device_noalloc, which works inside the library's static device array (noalloc.hip): with all three
// at their maximum the path pool takes 5.5 MB and one sample of radiance plus the sums 7.3 MB of the 16 MiB.
constexpr uint32_t kNoAllocSlots = 16384;
constexpr uint32_t kNoAllocMaxPixels = 512 * 512;
constexpr uint32_t kNoAllocMaxDepth = 64;

// The pieces of device memory one render call works in: the path pool, its bookkeeping, the framebuffer planes.
enum WsPiece { WS_RAYA, WS_RAYB, WS_RECENT, WS_RNG, WS_DEPTH, WS_HIT, WS_TRAIL, WS_QUEUE, WS_QCOUNT, WS_WAVE_STATS, WS_CURSORS,
               WS_FLAGS, WS_STATS_OUT, WS_ACCUM, WS_RAD, WS_PIXEL_MAP, WS_MERGED_QUEUE, WS_MERGED_QCOUNT, WS_LIST_OFFSETS, WS_COUNT };

struct RenderPlan {
    uint32_t npix = 0, batch = 0, slots = 0, n_waves = 0, trail_levels = 1;
    bool want_map = false;
    bool want_merge = false;  // a second list plane for the lists merged before the finishing launch
    size_t bytes[WS_COUNT] = {};
    // fills bytes[] from the sizes above; returns the total with every piece rounded up to 256 bytes
    size_t size_pieces() {
        const size_t sl = slots;
        bytes[WS_RAYA] = bytes[WS_RAYB] = bytes[WS_RECENT] = bytes[WS_RNG] = sl * 16;
        bytes[WS_DEPTH] = sl * 4; bytes[WS_HIT] = sl * 8; bytes[WS_TRAIL] = sl * 4 * trail_levels; bytes[WS_QUEUE] = sl * 4;
        bytes[WS_QCOUNT] = (size_t)n_waves * 4; bytes[WS_WAVE_STATS] = (size_t)n_waves * kWaveStatWords * 8;
        bytes[WS_CURSORS] = (size_t)kCursorShards * kCursorStride * 4; bytes[WS_FLAGS] = kFlagWords * 4; bytes[WS_STATS_OUT] = kWaveStatWords * 8;
        bytes[WS_ACCUM] = (size_t)npix * 12; bytes[WS_RAD] = (size_t)batch * npix * 16; bytes[WS_PIXEL_MAP] = want_map ? (size_t)npix * 4 : 0;
        bytes[WS_MERGED_QUEUE] = want_merge ? sl * 4 : 0; bytes[WS_MERGED_QCOUNT] = want_merge ? (size_t)n_waves * 4 : 0;
        bytes[WS_LIST_OFFSETS] = want_merge ? ((size_t)n_waves + 1) * 4 : 0;
        size_t total = 0;
        for (size_t b : bytes) total += (b + 255) & ~size_t(255);
        return total;
    }
};

}  // namespace

struct rrt_ctx {
    int device = 0;
    hipStream_t stream = nullptr, side_stream[kMaxParts - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxParts - 1] = {};
    hipDeviceProp_t prop{};
    // scene
    DevBuf nodes, small_nodes, top_nodes, wide, sph, motion, motion_dt, sph_info, prim_att;
    SceneDev sd{};
    bool has_scene = false, moving = false;
    uint32_t n_spheres = 0, n_nodes = 0;
    CameraDev cam{};
    bool has_cam = false;
    // the render workspace (one buffer per piece, grown on demand) and staging for the host-buffer entry points
    DevBuf ws[WS_COUNT];
    DevBuf tmp_rgb, tmp_acc;
    void *scratch = nullptr;       // this device's copy of the library's static workspace (rrt_render_device_noalloc)
    size_t scratch_bytes = 0;
    uint32_t *h_flags = nullptr;   // pinned: kFlagWords alive flags, then the path cursors
    uint64_t *h_stats = nullptr;   // pinned
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> kev;   // event pairs around extend launches
    // scratch for rrt_intersect
    DevBuf q_org, q_dir, q_time, q_t, q_obj;
    // the pixel list of the last TILE shard rendered, kept on the device: a rank renders the same shard frame after frame
    DevBuf tile_map;
    uint32_t tile_map_key[4] = {0, 0, 0, 0};  // nx, ny, rank, world (nx == 0: none)
    uint32_t tile_map_len = 0;
};

extern "C" {

const char *rrt_last_error(void) { return g_err.c_str(); }
const char *rrt_version(void) { return "rrt_amd 0.1 (gfx950)"; }

namespace {
bool make_side_streams(rrt_ctx *c) {
    for (uint32_t i = 0; i + 1 < kMaxParts; i++)
        if (hipStreamCreateWithFlags(&c->side_stream[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) != hipSuccess) return false;
    return true;
}
}  // namespace

int rrt_create(int device, rrt_ctx **out) {
    if (!out) return fail(RRT_ERR_INVALID_ARGUMENT, "rrt_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RRT_ERR_NO_DEVICE, "no HIP device available (there is no CPU fallback)");
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= n) return fail(RRT_ERR_INVALID_ARGUMENT, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    rrt_ctx *c = new rrt_ctx();
    c->device = device;
    if (hipGetDeviceProperties(&c->prop, device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || !make_side_streams(c) ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_flags), sizeof(uint32_t) * (kFlagWords + kCursorShards * kCursorStride), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&c->h_stats), sizeof(uint64_t) * kWaveStatWords, hipHostMallocDefault) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        rrt_destroy(c);
        return fail(RRT_ERR_HIP, "rrt_create: device set-up failed");
    }
    std::string err;
    if (rrt_amd::static_workspace(device, &c->scratch, &c->scratch_bytes, err) != RRT_OK) {
        rrt_destroy(c);
        return fail(RRT_ERR_HIP, "rrt_create: " + err);
    }
    // every stream a render may launch on gets its scratch memory sized here, where allocating is allowed (kernels.h)
    launch_prime_scratch(c->stream, static_cast<uint32_t *>(c->scratch));
    for (hipStream_t ss : c->side_stream) launch_prime_scratch(ss, static_cast<uint32_t *>(c->scratch));
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        rrt_destroy(c);
        return fail(RRT_ERR_HIP, "rrt_create: the first kernel launch on the device failed");
    }
    *out = c;
    return RRT_OK;
}

void rrt_destroy(rrt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipStream_t ss : c->side_stream) if (ss) (void)hipStreamSynchronize(ss);
    for (DevBuf *b : {&c->nodes, &c->small_nodes, &c->top_nodes, &c->wide, &c->sph, &c->motion, &c->motion_dt, &c->sph_info, &c->prim_att,
                      &c->tmp_rgb, &c->tmp_acc, &c->q_org, &c->q_dir, &c->q_time, &c->q_t, &c->q_obj, &c->tile_map})
        b->release();
    for (DevBuf &b : c->ws) b.release();
    for (hipEvent_t ev : c->kev) (void)hipEventDestroy(ev);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (hipEvent_t ev : c->ev_join) if (ev) (void)hipEventDestroy(ev);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipStream_t ss : c->side_stream) if (ss) (void)hipStreamDestroy(ss);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int rrt_device_info(rrt_ctx *c, char *name, size_t name_len, int *cus, uint64_t *hbm) {
    if (!c) return fail(RRT_ERR_INVALID_ARGUMENT, "null context");
    if (name && name_len) { std::snprintf(name, name_len, "%s (%s)", c->prop.name, c->prop.gcnArchName); }
    if (cus) *cus = c->prop.multiProcessorCount;
    if (hbm) *hbm = c->prop.totalGlobalMem;
    return RRT_OK;
}

namespace {
int upload_scene_impl(rrt_ctx *c, const rrt_sphere_soa *sp, const rrt_material_soa *mt, const rrt_bvh_node *nodes, uint32_t n_nodes);
}

int rrt_upload_scene(rrt_ctx *c, const rrt_sphere_soa *sp, const rrt_material_soa *mt, const rrt_bvh_node *nodes, uint32_t n_nodes) {
    if (!c) return fail(RRT_ERR_INVALID_ARGUMENT, "null context");
    try {
        return upload_scene_impl(c, sp, mt, nodes, n_nodes);
    } catch (const std::exception &e) {  // bad_alloc / system_error of the host-side layout pass
        return fail(RRT_ERR_OUT_OF_MEMORY, std::string("rrt_upload_scene: ") + e.what());
    }
}

namespace {
int upload_scene_impl(rrt_ctx *c, const rrt_sphere_soa *sp, const rrt_material_soa *mt, const rrt_bvh_node *nodes, uint32_t n_nodes) {
    SceneLayout L;
    std::string err;
    int rc = build_scene_layout(sp, mt, nodes, n_nodes, L, err);
    if (rc != RRT_OK) return fail(rc, err);
    make_top_tree(L);
    make_wide_nodes(L);
    HIP_TRY(hipSetDevice(c->device));
    c->has_scene = false;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(bytes ? bytes : 16);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    };
    HIP_TRY(up(c->nodes, L.nodes.data(), L.nodes.size() * 4));
    HIP_TRY(up(c->small_nodes, L.small_nodes.data(), L.small_nodes.size() * 4));
    HIP_TRY(up(c->top_nodes, L.top_nodes.data(), L.top_nodes.size() * 4));
    HIP_TRY(up(c->wide, L.wide.data(), L.wide.size() * 4));
    HIP_TRY(up(c->sph, L.sph.data(), L.sph.size() * sizeof(Float4)));
    HIP_TRY(up(c->motion, L.motion.data(), L.motion.size() * sizeof(Float4)));
    HIP_TRY(up(c->motion_dt, L.motion_dt.data(), L.motion_dt.size() * 4));
    HIP_TRY(up(c->sph_info, L.sph_info.data(), L.sph_info.size() * 4));
    HIP_TRY(up(c->prim_att, L.prim_att.data(), L.prim_att.size() * sizeof(Float4)));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->sd.nodes = c->nodes.as<uint4>();
    c->sd.root_ref = L.root_ref;
    c->sd.sph = c->sph.as<float4>();
    c->sd.motion = c->motion.as<float4>();
    c->sd.motion_dt = c->motion_dt.as<float>();
    c->sd.sph_info = c->sph_info.as<uint32_t>();
    c->sd.prim_att = c->prim_att.as<float4>();
    c->sd.flags = L.flags;
    c->sd.u_time0 = L.u_time0;
    c->sd.u_dt = L.u_dt;
    c->sd.small_nodes = L.small_nodes.empty() ? nullptr : c->small_nodes.as<uint4>();
    c->sd.small_bytes = L.small_bytes;
    c->sd.small_root = L.small_root;
    c->sd.small_end = L.small_end;
    c->sd.small_leaf_links = L.small_leaf_links;
    c->sd.n_nodes = L.n_nodes;
    c->sd.n_spheres = L.n_spheres;
    c->sd.top_nodes = L.top_nodes.empty() ? nullptr : c->top_nodes.as<uint4>();
    c->sd.n_top = (uint32_t)(L.top_nodes.size() / kNodeWords);
    c->sd.wide = L.wide.empty() ? nullptr : c->wide.as<uint4>();
    c->sd.wide_root = L.wide_root;
    for (int k = 0; k < 6; k++) c->sd.root_box[k] = L.root_box[k];
    c->moving = (L.flags & kSceneHasMoving) != 0;
    c->n_spheres = L.n_spheres;
    c->n_nodes = L.n_nodes;
    c->has_scene = true;
    return RRT_OK;
}
}  // namespace

int rrt_set_camera(rrt_ctx *c, const rrt_camera *cam) {
    if (!c || !cam) return fail(RRT_ERR_INVALID_ARGUMENT, "null argument");
    if (!std::isfinite(cam->time_low) || !std::isfinite(cam->time_scale))
        return fail(RRT_ERR_INVALID_ARGUMENT, "camera shutter times must be finite");
    for (int k = 0; k < 3; k++) {
        c->cam.llc[k] = cam->lower_left_corner[k]; c->cam.hor[k] = cam->horizontal[k]; c->cam.ver[k] = cam->vertical[k];
        c->cam.org[k] = cam->origin[k]; c->cam.u[k] = cam->u[k]; c->cam.v[k] = cam->v[k];
    }
    c->cam.lens_radius = cam->lens_radius;
    c->cam.time_low = cam->time_low;
    c->cam.time_scale = cam->time_scale;
    c->has_cam = true;
    return RRT_OK;
}

}  // extern "C"

namespace {

// Pixels of the tiles owned by `rank`, tile by tile (32x8), row-major inside a tile.
void tile_pixels(uint32_t nx, uint32_t ny, uint32_t rank, uint32_t world, std::vector<uint32_t> &out) {
    const uint32_t TW = 32, TH = 8;
    const uint32_t tx_n = (nx + TW - 1) / TW, ty_n = (ny + TH - 1) / TH;
    out.clear();
    for (uint32_t t = rank; t < tx_n * ty_n; t += world) {
        const uint32_t tx = t % tx_n, ty = t / tx_n;
        for (uint32_t yy = ty * TH; yy < std::min(ny, (ty + 1) * TH); yy++)
            for (uint32_t xx = tx * TW; xx < std::min(nx, (tx + 1) * TW); xx++) out.push_back(yy * nx + xx);
    }
}

// What the caller of render_impl supplies besides the context and the parameters.
struct RenderTarget {
    void *d_rgb8 = nullptr, *d_accum = nullptr;      // device memory of the caller (nullable)
    rrt_stats *stats = nullptr;
    volatile uint64_t *progress = nullptr;           // finished pixels, updated while the call runs (lib.rs:109)
    bool packed = false;                             // TILE shards: outputs hold only this rank's pixels, in rrt_shard_pixels order
    void *workspace = nullptr;                       // null: the context's own buffers, grown on demand; else: carve everything
    size_t workspace_bytes = 0;                      // out of this memory and allocate nothing
};

// The progress word is read by another thread while the render runs (main.rs:78-94 polls `pdc`): relaxed atomic accesses,
// as the reference's fetch_add(1, Relaxed) / load(Relaxed).
uint64_t progress_now(volatile uint64_t *p) { return __atomic_load_n(const_cast<uint64_t *>(p), __ATOMIC_RELAXED); }
void progress_set(volatile uint64_t *p, uint64_t v) { __atomic_store_n(const_cast<uint64_t *>(p), v, __ATOMIC_RELAXED); }

// Waits for everything the context has queued, on every stream.  Error paths call it so that no kernel of a failed
// render is still running on a side stream when the caller's next call reuses the buffers.
void drain(rrt_ctx *c) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipStream_t ss : c->side_stream) if (ss) (void)hipStreamSynchronize(ss);
}

int render_queued(rrt_ctx *c, const rrt_render_params &P, const RenderTarget &T, rrt_stats &st, size_t &kev_used, uint32_t &n_waves_out);

int render_impl(rrt_ctx *c, const rrt_render_params *pp, const RenderTarget &T) {
    if (!c || !pp) return fail(RRT_ERR_INVALID_ARGUMENT, "null argument");
    if (!c->has_scene || !c->has_cam) return fail(RRT_ERR_NOT_READY, "render before rrt_upload_scene / rrt_set_camera");
    const rrt_render_params P = *pp;
    if (P.nx == 0 || P.ny == 0 || P.ns == 0) return fail(RRT_ERR_INVALID_ARGUMENT, "nx, ny and ns must be positive");
    if ((uint64_t)P.nx * P.ny > (1ull << 28)) return fail(RRT_ERR_INVALID_ARGUMENT, "image too large");
    if (P.max_depth > kMaxDepthLimit) return fail(RRT_ERR_INVALID_ARGUMENT, "max_depth too large");
    const uint32_t world = P.shard_mode == RRT_SHARD_NONE ? 1u : P.shard_world;
    const uint32_t rank = P.shard_mode == RRT_SHARD_NONE ? 0u : P.shard_rank;
    if (P.shard_mode > RRT_SHARD_SAMPLE || world == 0 || rank >= world) return fail(RRT_ERR_INVALID_ARGUMENT, "bad shard description");
    HIP_TRY(hipSetDevice(c->device));
    if (T.progress) progress_set(T.progress, 0);
    rrt_stats st{};
    size_t kev_used = 0;
    uint32_t n_waves = 0;
    const int rc = render_queued(c, P, T, st, kev_used, n_waves);
    if (rc != RRT_OK) {
        const std::string msg = g_err;
        drain(c);
        (void)hipGetLastError();
        g_err = msg;
        return rc;
    }
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    st.gpu_ms = ms;
    if (n_waves) {
        const uint64_t *h = c->h_stats;
        st.rays = h[ST_RAYS]; st.inner_visits = h[ST_INNER]; st.leaf_tests = h[ST_LEAF];
        st.hits_lambertian = h[ST_HL]; st.hits_metal = h[ST_HM]; st.hits_dielectric = h[ST_HD];
        st.sky_misses = h[ST_SKY]; st.paths = h[ST_PATHS]; st.depth_cut = h[ST_DEPTHCUT]; st.absorbed = h[ST_ABSORBED];
        for (int k = 0; k < 6; k++) st.debug[k] = h[10 + k];
    }
    double ext = 0.0;
    for (size_t i = 0; i + 1 < kev_used; i += 2) {
        float m = 0.f;
        if (hipEventElapsedTime(&m, c->kev[i], c->kev[i + 1]) == hipSuccess) ext += m;
    }
    st.extend_ms = ext;
    st.extend_launches = kev_used / 2;
    if (T.stats) *T.stats = st;
    return RRT_OK;
}

// Everything of a render call that queues work; returns with all streams idle on success.
int render_queued(rrt_ctx *c, const rrt_render_params &P, const RenderTarget &T, rrt_stats &st, size_t &kev_used, uint32_t &n_waves_out) {
    const bool no_alloc = T.workspace != nullptr;
    const uint32_t world = P.shard_mode == RRT_SHARD_NONE ? 1u : P.shard_world;
    const uint32_t rank = P.shard_mode == RRT_SHARD_NONE ? 0u : P.shard_rank;
    hipStream_t s = c->stream;
    const uint32_t npix_img = P.nx * P.ny;
    void *const d_rgb8 = T.d_rgb8, *const d_accum = T.d_accum;

    // ---- which pixels and which samples this call renders ----
    RenderPlan pl;
    pl.npix = npix_img;
    const uint32_t *d_map = nullptr;
    if (P.shard_mode == RRT_SHARD_TILE && world > 1) {
        if (no_alloc) return fail(RRT_ERR_INVALID_ARGUMENT, "the no-allocation entry point takes RRT_SHARD_NONE or RRT_SHARD_SAMPLE only");
        const uint32_t key[4] = {P.nx, P.ny, rank, world};
        if (std::memcmp(key, c->tile_map_key, sizeof key) != 0) {  // another shard than last time: list its pixels and upload them
            std::vector<uint32_t> map;
            tile_pixels(P.nx, P.ny, rank, world, map);
            c->tile_map_key[0] = 0;
            HIP_TRY(c->tile_map.ensure(std::max<size_t>(16, map.size() * 4)));
            HIP_TRY(hipMemcpyAsync(c->tile_map.p, map.data(), map.size() * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));  // pageable source
            c->tile_map_len = (uint32_t)map.size();
            std::memcpy(c->tile_map_key, key, sizeof key);
        }
        pl.npix = c->tile_map_len;
        d_map = c->tile_map.as<uint32_t>();
    }
    const uint32_t npix = pl.npix;
    uint32_t k_first = 0, k_step = 1, ns_local = P.ns;
    if (P.shard_mode == RRT_SHARD_SAMPLE && world > 1) {
        k_first = rank; k_step = world;
        ns_local = rank < P.ns ? (P.ns - rank + world - 1) / world : 0;
    }

    const size_t npix_out = T.packed ? npix : npix_img;
    if (d_rgb8) HIP_TRY(hipMemsetAsync(d_rgb8, 0, npix_out * 3, s));
    if (d_accum) HIP_TRY(hipMemsetAsync(d_accum, 0, npix_out * 12, s));
    HIP_TRY(hipEventRecord(c->ev0, s));

    const bool count = P.count_work != 0;
    const uint32_t retire_min = P.reserved[2] ? std::min<uint32_t>(P.reserved[2], 64u) : 40u;  // tuning knob of the extend kernel
    const uint32_t leaf_min = P.reserved[3] ? std::min<uint32_t>(P.reserved[3], 64u) : 16u;
    SceneDev sd = c->sd;
    if (P.reserved[1] & 1u) sd.small_nodes = nullptr;  // walk the HBM copy of the BVH even if the scene fits in LDS
    // reserved[1] bit 1: walk big scenes node by node along the exit links instead of two children per record.  The entry
    // point that allocates nothing does so always: the two-children walk keeps a stack in scratch memory, which the
    // runtime allocates when such a kernel is first launched.
    if ((P.reserved[1] & 2u) || no_alloc || std::getenv("RRT_NO_WIDE")) sd.wide = nullptr;
    // reserved[0] = n: HIP events around every n-th extend launch of each part (n divides the rounds of a chunk, so the
    // timed launches are an even 1-in-n sample); the counting build times every launch.  Timing creates events on
    // demand, so the entry point that must not allocate leaves it out.
    uint32_t time_stride = count ? 1u : (P.reserved[0] <= kChunk && P.reserved[0] != 0 && kChunk % P.reserved[0] == 0 ? P.reserved[0] : (P.reserved[0] ? 1u : 0u));
    if (no_alloc) time_stride = 0;
    if (npix > 0 && ns_local > 0) {
        // ---- batch and slot sizing ----
        pl.trail_levels = std::max<uint32_t>(1, P.max_depth);
        uint32_t slots = P.path_slots ? P.path_slots : (no_alloc ? kNoAllocSlots : kDefaultSlots);
        if (no_alloc) {
            if (npix > kNoAllocMaxPixels || pl.trail_levels > kNoAllocMaxDepth || slots > kNoAllocSlots)
                return fail(RRT_ERR_INVALID_ARGUMENT, "the no-allocation entry point takes at most " + std::to_string(kNoAllocMaxPixels) +
                            " pixels, max_depth " + std::to_string(kNoAllocMaxDepth) + " and " + std::to_string(kNoAllocSlots) + " path slots");
        }
        size_t rad_budget = kRadBudgetBytes;
        if (!no_alloc) {  // leave room for whatever else lives on the device
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) rad_budget = std::min(rad_budget, std::max<size_t>(free_b / 2 + c->ws[WS_RAD].cap, size_t(64) << 20));
        }
        uint32_t batch = P.batch_samples ? P.batch_samples : (uint32_t)std::min<size_t>(0xFFFFFFFFu, std::max<size_t>(1, rad_budget / ((size_t)npix * 16)));
        batch = std::min(batch, ns_local);
        batch = std::min<uint32_t>(batch, std::max<uint32_t>(1, kMaxBatchPaths / npix));
        slots = std::min<uint64_t>(slots, ((uint64_t)batch * npix + kWaveSlots - 1) / kWaveSlots * kWaveSlots);
        slots = std::max(kWaveSlots, (slots + kWaveSlots - 1) / kWaveSlots * kWaveSlots);
        pl.slots = slots;
        pl.n_waves = slots / kWaveSlots;
        pl.batch = batch;
        pl.want_merge = !no_alloc;
        void *piece[WS_COUNT] = {};
        if (no_alloc) {
            // as many samples per batch as the workspace holds beside the path pool
            pl.batch = 0;
            const size_t fixed = pl.size_pieces();
            if (fixed + (((size_t)npix * 16 + 255) & ~size_t(255)) > T.workspace_bytes) return fail(RRT_ERR_INVALID_ARGUMENT, "workspace too small");
            pl.batch = (uint32_t)std::min<size_t>(batch, (T.workspace_bytes - fixed - 256) / ((size_t)npix * 16));
            pl.size_pieces();
            char *at = static_cast<char *>(T.workspace);
            for (int i = 0; i < WS_COUNT; i++) { piece[i] = at; at += (pl.bytes[i] + 255) & ~size_t(255); }
        } else {
            pl.size_pieces();
            for (int i = 0; i < WS_COUNT; i++) { HIP_TRY(c->ws[i].ensure(std::max<size_t>(16, pl.bytes[i]))); piece[i] = c->ws[i].p; }
        }
        batch = pl.batch;
        const uint32_t n_waves = pl.n_waves;
        n_waves_out = n_waves;
        auto at = [&](int i) { return piece[i]; };

        PathState ps{};
        ps.rayA = static_cast<float4 *>(at(WS_RAYA)); ps.rayB = static_cast<float4 *>(at(WS_RAYB));
        ps.recent = static_cast<uint4 *>(at(WS_RECENT)); ps.rng = static_cast<uint4 *>(at(WS_RNG));
        ps.depth = static_cast<uint32_t *>(at(WS_DEPTH)); ps.hit = static_cast<uint2 *>(at(WS_HIT)); ps.trail = static_cast<uint32_t *>(at(WS_TRAIL));
        ps.queue = static_cast<uint32_t *>(at(WS_QUEUE)); ps.qcount = static_cast<uint32_t *>(at(WS_QCOUNT)); ps.n_slots = slots; ps.trail_levels = pl.trail_levels;
        ps.wave_begin = 0; ps.wave_end = n_waves;
        ps.t_min = 0.001f; ps.t_max = std::numeric_limits<float>::max();  // lib.rs:57
        ps.layout = (c->moving ? kLayoutTimed : 0u) | (c->n_spheres <= kShadeTableSpheres && P.max_depth <= kNarrowMaxDepth ? kLayoutNarrow : 0u);
        {   // the extend kernel that walks the tree in LDS takes the rays of freshly started paths last (RRT_FRESH_APART=0: slot
            // order).  Not the walks in HBM: the 1 M-sphere scene is 3 % slower with it.
            const char *env_fa = std::getenv("RRT_FRESH_APART");
            if (sd.small_nodes && !(env_fa && env_fa[0] == '0')) ps.layout |= kLayoutFreshApart;
        }
        uint64_t *wstats = static_cast<uint64_t *>(at(WS_WAVE_STATS));
        HIP_TRY(hipMemsetAsync(wstats, 0, (size_t)n_waves * kWaveStatWords * 8, s));
        uint32_t *flags = static_cast<uint32_t *>(at(WS_FLAGS));
        uint32_t *cursors = static_cast<uint32_t *>(at(WS_CURSORS));
        float *accum = static_cast<float *>(at(WS_ACCUM));
        float4 *rad = static_cast<float4 *>(at(WS_RAD));

        // The path pool is worked on in parts, each with its own stream running extend, shade, extend, ... freely, so
        // that the extend kernel of one part (VALU-bound) runs beside the shade kernel of another (bound by memory
        // latency).  RRT_STREAMS=1..4 picks the number of parts.
        const char *env_streams = std::getenv("RRT_STREAMS");
        uint32_t n_parts = sd.small_nodes ? (c->moving ? kDefaultPartsLdsMoving : kDefaultPartsLds) : kDefaultPartsHbm;
        if (env_streams && env_streams[0] >= '1' && env_streams[0] <= '0' + (int)kMaxParts) n_parts = (uint32_t)(env_streams[0] - '0');
        if (n_waves < 32 * n_parts) n_parts = 1;
        PathState part[kMaxParts];
        hipStream_t part_stream[kMaxParts];
        for (uint32_t h = 0; h < n_parts; h++) {
            part[h] = ps;
            part[h].wave_begin = (uint32_t)((uint64_t)n_waves * h / n_parts);
            part[h].wave_end = (uint32_t)((uint64_t)n_waves * (h + 1) / n_parts);
            part_stream[h] = h == 0 ? s : c->side_stream[h - 1];
        }
            uint32_t chunk = kChunk, rounds_done = 0;
        // (not for the entry point that must allocate nothing: the finishing kernel spills a few registers, and the runtime
        // allocates the memory spills go to when such a kernel is first launched)
        const bool can_finish = !no_alloc && finish_fits(sd, ps) && !std::getenv("RRT_NO_FINISH");
        // Between the last path being dealt and the finishing launch: a few more plain rounds (what is left of a full pool is
        // still best served by the two kernels), then the thinned-out lists are merged into full ones for the first waves
        // (kernels.h: launch_merge_lists), so that the finishing launch is a fraction of the pool's workgroups, all at work,
        // instead of every workgroup of the pool waiting for its longest path.  RRT_FINISH_MERGE=0: the finishing launch at
        // once, on the lists as they are (2: merge whatever the size of the pool); RRT_FINISH_DELAY: the number of rounds.
        uint32_t finish_delay = kFinishDelayRounds;
        if (const char *e = std::getenv("RRT_FINISH_DELAY")) finish_delay = (uint32_t)std::min(64, std::max(0, std::atoi(e)));
        bool merge_lists = pl.want_merge && n_waves >= kMergeMinWaves;
        if (const char *e = std::getenv("RRT_FINISH_MERGE")) merge_lists = pl.want_merge && (e[0] == '2' || (merge_lists && e[0] != '0'));  // 2: pools of any size
        PathState ps_finish = ps;  // what the finishing launch works on
        for (uint32_t s0 = 0; s0 < ns_local; s0 += batch) {
            const uint32_t nb = std::min(batch, ns_local - s0);
            BatchDev bd{};
            bd.nx = P.nx; bd.ny = P.ny; bd.npix = npix; bd.pixel_map = d_map;
            bd.div_nx = make_fastdiv(P.nx); bd.div_npix = make_fastdiv(npix);
            bd.inv_nx = 1.0f / (float)P.nx; bd.inv_ny = 1.0f / (float)P.ny;
            bd.n_paths = nb * npix;
            bd.k_first = k_first + s0 * k_step; bd.k_step = k_step;
            bd.seed = P.seed;
            bd.rad = rad;
            bd.max_depth = P.max_depth;
            bd.cursors = cursors;
            bd.n_shards = std::min(kCursorShards, n_waves);
            bd.alive_flag = nullptr;  // raygen's survivors are not looked at
            HIP_TRY(hipMemsetAsync(bd.cursors, 0, (size_t)kCursorShards * kCursorStride * 4, s));
            launch_raygen(s, ps, bd, c->cam, wstats);
            // Once every path of the batch has been dealt (at once, if the batch is no larger than the pool), what is left
            // thins out over up to max_depth more rounds: where the scene allows it, one launch runs them all.
            bool dealt = can_finish && (uint64_t)bd.n_paths <= slots;
            bool merge_pending = false;  // the next finishing launch is preceded by the plain rounds and the merge
            ps_finish = ps;
            uint64_t claimed_before = std::min<uint64_t>(slots, bd.n_paths);  // raygen deals the first paths
            for (;;) {
                HIP_TRY(hipMemsetAsync(flags, 0, kFlagWords * 4, s));
                // `rounds` rounds of extend + shade on every part's stream, joined on `s`; the last round reports which waves still
                // hold paths if `report` is set
                auto queue_rounds = [&](uint32_t rounds, bool report) -> int {
                    if (n_parts > 1) {
                        HIP_TRY(hipEventRecord(c->ev_fork, s));
                        for (uint32_t h = 1; h < n_parts; h++) HIP_TRY(hipStreamWaitEvent(part_stream[h], c->ev_fork, 0));
                    }
                    for (uint32_t it = 0; it < rounds; it++) {
                        for (uint32_t h = 0; h < n_parts; h++) {
                            hipStream_t sp = part_stream[h];
                            bool time_kernels = time_stride != 0 && it % time_stride == 0;
                            if (time_kernels) {
                                while (c->kev.size() < kev_used + 2) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); c->kev.push_back(ev); }
                                HIP_TRY(hipEventRecord(c->kev[kev_used], sp));
                            }
                            launch_extend(sp, sd, part[h], c->moving, count, retire_min, leaf_min, wstats);
                            if (time_kernels) { HIP_TRY(hipEventRecord(c->kev[kev_used + 1], sp)); kev_used += 2; }
                            bd.alive_flag = report && it + 1 == rounds ? flags + h * 1024 : nullptr;  // only the chunk's last round reports
                            launch_shade(sp, sd, part[h], bd, c->cam, c->moving, count, wstats);
                        }
                    }
                    st.iterations += rounds;
                    for (uint32_t h = 1; h < n_parts; h++) { HIP_TRY(hipEventRecord(c->ev_join[h - 1], part_stream[h])); HIP_TRY(hipStreamWaitEvent(s, c->ev_join[h - 1], 0)); }
                    return RRT_OK;
                };
                if (dealt) {
                    if (merge_pending) {
                        merge_pending = false;
                        if (finish_delay) { const int rc = queue_rounds(finish_delay, false); if (rc != RRT_OK) return rc; }
                        ps_finish.queue = static_cast<uint32_t *>(at(WS_MERGED_QUEUE));
                        ps_finish.qcount = static_cast<uint32_t *>(at(WS_MERGED_QCOUNT));
                        launch_merge_lists(s, ps, ps_finish.queue, ps_finish.qcount, static_cast<uint32_t *>(at(WS_LIST_OFFSETS)), kMergedListLength);
                    }
                    if (launch_finish(s, sd, ps_finish, bd, c->cam, c->moving, count, P.max_depth + 2u, retire_min, leaf_min, flags, wstats))
                        return fail(RRT_ERR_HIP, "rrt_render: the finishing kernel's LDS request was refused");
                    st.iterations += 1;
                } else {
                    const int rc = queue_rounds(chunk, true);
                    if (rc != RRT_OK) return rc;
                    rounds_done = chunk;
                    chunk = std::min(kMaxChunk, chunk * 2u);
                }
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(c->h_flags, flags, kFlagWords * 4, hipMemcpyDeviceToHost, s));
                if (T.progress || (can_finish && !dealt))
                    HIP_TRY(hipMemcpyAsync(c->h_flags + kFlagWords, cursors, (size_t)kCursorShards * kCursorStride * 4, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                uint32_t any = 0;
                for (uint32_t k = 0; k < kFlagWords; k += 32) any |= c->h_flags[k];
                if (can_finish && !dealt) {  // dealt: the next claim of every cursor shard lies past the batch (kernels.hip: path_of_claim)
                    dealt = true;
                    uint64_t claimed = 0;
                    for (uint32_t k = 0; k < bd.n_shards; k++) {
                        const uint64_t n = c->h_flags[kFlagWords + k * kCursorStride];
                        claimed += n;
                        if (((n >> 6) * bd.n_shards + k) * 64ull + (n & 63u) < (uint64_t)bd.n_paths) dealt = false;
                    }
                    // The rounds after the last path is dealt belong to the finishing launch: the next chunk is no longer than
                    // the rounds the batch's remaining paths last at the rate the last chunk took them.
                    claimed = std::min<uint64_t>(claimed, bd.n_paths);
                    if (!dealt && rounds_done != 0 && claimed > claimed_before) {
                        const double per_round = (double)(claimed - claimed_before) / rounds_done;
                        const double left = (double)(bd.n_paths - claimed) / per_round;
                        if (left < (double)chunk) chunk = std::max(1u, (uint32_t)std::ceil(left));
                    }
                    claimed_before = claimed;
                    if (dealt && merge_lists && any) merge_pending = true;
                }
                if (T.progress) {
                    // Paths handed out so far, less those that may still be in flight, are finished; a pixel counts as
                    // finished in proportion to its finished samples (the reference bumps its counter per pixel, lib.rs:109)
                    uint64_t claimed = 0;
                    for (uint32_t k = 0; k < kCursorShards; k++) claimed += c->h_flags[kFlagWords + k * kCursorStride];
                    claimed = std::min<uint64_t>(claimed, bd.n_paths);
                    const uint64_t finished = (uint64_t)s0 * npix + (any ? (claimed > slots ? claimed - slots : 0) : bd.n_paths);
                    const uint64_t px = finished / ns_local;  // of total_paths = ns_local * npix paths: finished * npix / total_paths
                    if (px > progress_now(T.progress)) progress_set(T.progress, px);
                }
                if (!any) break;
            }
            launch_accumulate(s, accum, npix, rad, nb, s0 == 0);
        }
        // a SAMPLE shard's own RGB8 is the image of its samples alone: divide by their number, not by the whole job's
        launch_finalize(s, accum, npix, T.packed ? nullptr : d_map, ns_local, static_cast<uint8_t *>(d_rgb8), static_cast<float *>(d_accum));
        launch_reduce_stats(s, wstats, n_waves, static_cast<uint64_t *>(at(WS_STATS_OUT)));
        HIP_TRY(hipMemcpyAsync(c->h_stats, at(WS_STATS_OUT), kWaveStatWords * 8, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipEventRecord(c->ev1, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    if (T.progress) progress_set(T.progress, npix);  // every pixel of this call is finished (lib.rs:109)
    return RRT_OK;
}

}  // namespace

extern "C" {

int rrt_build_bvh(rrt_ctx *c, const rrt_sphere_soa *sp, rrt_bvh_node *out_nodes, uint32_t capacity, uint32_t *n_nodes) {
    return rrt_build_bvh_method(c, sp, RRT_BVH_MIDDLE, out_nodes, capacity, n_nodes);
}

int rrt_build_bvh_method(rrt_ctx *c, const rrt_sphere_soa *sp, uint32_t method, rrt_bvh_node *out_nodes, uint32_t capacity, uint32_t *n_nodes) {
    if (!c || !sp || !out_nodes || !n_nodes) return fail(RRT_ERR_INVALID_ARGUMENT, "rrt_build_bvh: null argument");
    if (method > RRT_BVH_EQUAL_COUNT) return fail(RRT_ERR_INVALID_ARGUMENT, "rrt_build_bvh: unknown partition method");
    if (sp->n == 0) return fail(RRT_ERR_BAD_SCENE, "rrt_build_bvh: no objects (Bvh::build of an empty scene yields no nodes)");
    if (sp->n > (kRefIndexMask + 1u) / 2u) return fail(RRT_ERR_BAD_SCENE, "rrt_build_bvh: too many objects");  // 2n-1 nodes must fit rrt_upload_scene
    if (!sp->c0x || !sp->c0y || !sp->c0z || !sp->c1x || !sp->c1y || !sp->c1z || !sp->radius)
        return fail(RRT_ERR_INVALID_ARGUMENT, "rrt_build_bvh: null sphere plane");
    HIP_TRY(hipSetDevice(c->device));
    try {
        std::string err;
        const int rc = rrt_amd::gpu_build_bvh(c->stream, sp, method, out_nodes, capacity, n_nodes, err);
        return rc == RRT_OK ? rc : fail(rc, err);
    } catch (const std::exception &e) { return fail(RRT_ERR_OUT_OF_MEMORY, std::string("rrt_build_bvh: ") + e.what()); }
}

int rrt_render_device(rrt_ctx *c, const rrt_render_params *p, void *d_rgb8, void *d_accum, rrt_stats *stats) {
    try {
        RenderTarget T;
        T.d_rgb8 = d_rgb8; T.d_accum = d_accum; T.stats = stats;
        return render_impl(c, p, T);
    } catch (const std::exception &e) { return fail(RRT_ERR_OUT_OF_MEMORY, std::string("rrt_render_device: ") + e.what()); }
}

int rrt_render(rrt_ctx *c, const rrt_render_params *p, uint8_t *out_rgb8, float *out_accum, rrt_stats *stats, volatile uint64_t *progress) {
    if (!c || !p) return fail(RRT_ERR_INVALID_ARGUMENT, "null argument");
    if (!out_rgb8) return fail(RRT_ERR_INVALID_ARGUMENT, "out_rgb8 is null");
    const size_t npix = (size_t)p->nx * p->ny;
    if (npix == 0) return fail(RRT_ERR_INVALID_ARGUMENT, "nx and ny must be positive");
    try {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(c->tmp_rgb.ensure(npix * 3));
        if (out_accum) HIP_TRY(c->tmp_acc.ensure(npix * 12));
        RenderTarget T;
        T.d_rgb8 = c->tmp_rgb.p; T.d_accum = out_accum ? c->tmp_acc.p : nullptr; T.stats = stats;
        T.progress = progress;
        int rc = render_impl(c, p, T);
        if (rc != RRT_OK) return rc;
        HIP_TRY(hipMemcpy(out_rgb8, c->tmp_rgb.p, npix * 3, hipMemcpyDeviceToHost));
        if (out_accum) HIP_TRY(hipMemcpy(out_accum, c->tmp_acc.p, npix * 12, hipMemcpyDeviceToHost));
        return RRT_OK;
    } catch (const std::exception &e) { return fail(RRT_ERR_OUT_OF_MEMORY, std::string("rrt_render: ") + e.what()); }
}

int rrt_validate_scene(const rrt_sphere_soa *sp, const rrt_material_soa *mt, const rrt_bvh_node *nodes, uint32_t n_nodes) {
    try {
        SceneLayout L;
        std::string err;
        int rc = build_scene_layout(sp, mt, nodes, n_nodes, L, err);
        if (rc != RRT_OK) return fail(rc, err);
        return RRT_OK;
    } catch (const std::exception &e) { return fail(RRT_ERR_OUT_OF_MEMORY, std::string("rrt_validate_scene: ") + e.what()); }
}

int rrt_debug_layout(const rrt_sphere_soa *sp, const rrt_material_soa *mt, const rrt_bvh_node *nodes, uint32_t n_nodes, uint32_t *words16,
                     uint32_t *root_ref, uint32_t *scene_flags) {
    try {
        SceneLayout L;
        std::string err;
        int rc = build_scene_layout(sp, mt, nodes, n_nodes, L, err);
        if (rc != RRT_OK) return fail(rc, err);
        if (words16) std::memcpy(words16, L.nodes.data(), L.nodes.size() * 4);
        if (root_ref) *root_ref = L.root_ref;
        if (scene_flags) {  // with the bits the later layout passes add (they rewrite references: after the copies above)
            make_top_tree(L);
            make_wide_nodes(L);
            *scene_flags = L.flags;
        }
        return RRT_OK;
    } catch (const std::exception &e) { return fail(RRT_ERR_OUT_OF_MEMORY, std::string("rrt_debug_layout: ") + e.what()); }
}

uint32_t rrt_shard_pixels(uint32_t nx, uint32_t ny, uint32_t shard_mode, uint32_t rank, uint32_t world, uint32_t *out_idx) {
    if (shard_mode == RRT_SHARD_TILE && world > 1) {
        if (rank >= world) return 0;
        std::vector<uint32_t> map;
        tile_pixels(nx, ny, rank, world, map);
        if (out_idx) std::memcpy(out_idx, map.data(), map.size() * 4);
        return (uint32_t)map.size();
    }
    const uint32_t n = nx * ny;
    if (out_idx) for (uint32_t i = 0; i < n; i++) out_idx[i] = i;
    return n;
}

int rrt_finalize(rrt_ctx *c, const void *d_accum, uint32_t n_pixels, uint32_t ns, void *d_rgb8) {
    if (!c || !d_accum || !d_rgb8 || ns == 0) return fail(RRT_ERR_INVALID_ARGUMENT, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (n_pixels) launch_finalize_interleaved(c->stream, static_cast<const float *>(d_accum), n_pixels, ns, static_cast<uint8_t *>(d_rgb8));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RRT_OK;
}

int rrt_intersect(rrt_ctx *c, uint32_t n, const float *org, const float *dir, const float *time, float t_min, float t_max, float *out_t,
                  uint32_t *out_obj) {
    if (!c || !org || !dir || !out_t || !out_obj) return fail(RRT_ERR_INVALID_ARGUMENT, "null argument");
    if (!c->has_scene) return fail(RRT_ERR_NOT_READY, "rrt_intersect before rrt_upload_scene");
    if (n == 0) return RRT_OK;
    if (n > (1u << 28)) return fail(RRT_ERR_INVALID_ARGUMENT, "too many rays");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIP_TRY(c->q_org.ensure((size_t)n * 12)); HIP_TRY(c->q_dir.ensure((size_t)n * 12)); HIP_TRY(c->q_time.ensure((size_t)n * 4));
    HIP_TRY(c->q_t.ensure((size_t)n * 4)); HIP_TRY(c->q_obj.ensure((size_t)n * 4));
    HIP_TRY(hipMemcpyAsync(c->q_org.p, org, (size_t)n * 12, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->q_dir.p, dir, (size_t)n * 12, hipMemcpyHostToDevice, s));
    if (time) HIP_TRY(hipMemcpyAsync(c->q_time.p, time, (size_t)n * 4, hipMemcpyHostToDevice, s));
    // the rays take the first n slots of the path pool, and the render loop's extend kernel walks them
    RenderPlan pl;
    pl.slots = (n + kWaveSlots - 1) / kWaveSlots * kWaveSlots;
    pl.n_waves = pl.slots / kWaveSlots;
    pl.size_pieces();
    for (int i : {WS_RAYA, WS_RAYB, WS_HIT, WS_QUEUE, WS_QCOUNT, WS_WAVE_STATS}) HIP_TRY(c->ws[i].ensure(std::max<size_t>(16, pl.bytes[i])));
    PathState ps{};
    ps.rayA = c->ws[WS_RAYA].as<float4>(); ps.rayB = c->ws[WS_RAYB].as<float4>();
    ps.hit = c->ws[WS_HIT].as<uint2>(); ps.queue = c->ws[WS_QUEUE].as<uint32_t>(); ps.qcount = c->ws[WS_QCOUNT].as<uint32_t>();
    ps.n_slots = pl.slots; ps.trail_levels = 1; ps.wave_begin = 0; ps.wave_end = pl.n_waves;
    ps.t_min = t_min; ps.t_max = t_max;
    ps.layout = c->moving ? kLayoutTimed : 0u;
    launch_intersect_setup(s, ps, n, c->q_org.as<float>(), c->q_dir.as<float>(), time ? c->q_time.as<float>() : nullptr);
    SceneDev sd = c->sd;
    if (!(t_min > 0.0f)) sd.wide = nullptr;  // the two-children walk keeps a mark in the sign of entry distances, which are t_min or more: along the exit links otherwise
    launch_extend(s, sd, ps, c->moving, false, 32u, 16u, c->ws[WS_WAVE_STATS].as<uint64_t>());
    launch_intersect_gather(s, ps, n, c->q_t.as<float>(), c->q_obj.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_t, c->q_t.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_obj, c->q_obj.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return RRT_OK;
}

}  // extern "C"

namespace rrt_amd {

int render_in_workspace(rrt_ctx *ctx, const rrt_render_params *params, void *d_rgb8, void *d_accum, rrt_stats *stats, void *workspace, size_t workspace_bytes) {
    RenderTarget T;
    T.d_rgb8 = d_rgb8; T.d_accum = d_accum; T.stats = stats;
    T.workspace = workspace; T.workspace_bytes = workspace_bytes;
    return render_impl(ctx, params, T);
}
int context_device(const rrt_ctx *ctx) { return ctx ? ctx->device : -1; }
hipStream_t context_stream(const rrt_ctx *ctx) { return ctx->stream; }
// for multi.cpp: one rank's share of a sharded render, results left on the context's device; TILE shards come out packed
int render_shard(rrt_ctx *ctx, const rrt_render_params *params, void *d_rgb8, void *d_accum, rrt_stats *stats, volatile uint64_t *progress, bool packed) {
    try {
        RenderTarget T;
        T.d_rgb8 = d_rgb8; T.d_accum = d_accum; T.stats = stats; T.progress = progress; T.packed = packed;
        return render_impl(ctx, params, T);
    } catch (const std::exception &e) { return fail(RRT_ERR_OUT_OF_MEMORY, std::string("render: ") + e.what()); }
}
void shard_pixel_list(uint32_t nx, uint32_t ny, uint32_t rank, uint32_t world, std::vector<uint32_t> &out) { tile_pixels(nx, ny, rank, world, out); }
const char *last_error_text() { return g_err.c_str(); }
int set_error(int code, const std::string &msg) { return fail(code, msg); }

}  // namespace rrt_amd
