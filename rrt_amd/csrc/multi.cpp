// rrt_render_multi: one call, whole image back (lib.rs:73-81), rendered on several GPUs of one node.
//
// One host thread per context renders that rank's shard (TILE: 32x8 tiles dealt round-robin; SAMPLE: sample k -> rank
// k mod n) and the framebuffers are combined on rank 0's GPU over RCCL:
//   TILE    every peer sends the RGB8 (and, if asked for, f32) pixels it owns, packed, straight to rank 0 — one ncclSend
//           per peer and plane inside one group, 24.9 MB of RGB8 in all at 3840x2160; rank 0 puts them in place;
//   SAMPLE  ncclReduce(sum) of the f32 sums onto rank 0, which quantises once (lib.rs:105-108).
// xGMI links every pair of GPUs directly, so one hop per peer is the whole exchange.  The work counters are summed on the
// host: all ranks live in this process.
//
// RCCL is loaded on first use (dlopen), so programs that render on one GPU never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rrt_amd.h"
#include "kernels.h"

namespace rrt_amd {
int context_device(const rrt_ctx *ctx);
hipStream_t context_stream(const rrt_ctx *ctx);
int render_shard(rrt_ctx *ctx, const rrt_render_params *params, void *d_rgb8, void *d_accum, rrt_stats *stats, volatile uint64_t *progress, bool packed);
void shard_pixel_list(uint32_t nx, uint32_t ny, uint32_t rank, uint32_t world, std::vector<uint32_t> &out);
int set_error(int code, const std::string &msg);
const char *last_error_text();
}  // namespace rrt_amd

using namespace rrt_amd;

namespace {

// ---- the few RCCL entry points used, resolved at run time (declarations as in rccl.h) ----
typedef struct ncclComm *ncclComm_t;
enum { kNcclSuccess = 0, kNcclUint8 = 1, kNcclFloat32 = 7, kNcclSum = 0 };
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load(std::string &err) {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        Reduce = reinterpret_cast<decltype(Reduce)>(sym("ncclReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !Reduce || !GetErrorString) {
            err = "librccl lacks an expected entry point";
            dlclose(lib); lib = nullptr;
            return false;
        }
        return true;
    }
};

// Communicators are expensive to make (seconds) and cheap to keep: one set per distinct device list, for the process's life.
struct CommSet { std::vector<int> devices; std::vector<ncclComm_t> comms; };
std::mutex g_mutex;
Rccl g_rccl;
std::vector<CommSet *> g_sets;

CommSet *comm_set_for(const std::vector<int> &devices, std::string &err) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_rccl.load(err)) return nullptr;
    for (CommSet *s : g_sets) if (s->devices == devices) return s;
    CommSet *s = new CommSet();
    s->devices = devices;
    s->comms.resize(devices.size());
    const int rc = g_rccl.CommInitAll(s->comms.data(), (int)devices.size(), devices.data());
    if (rc != kNcclSuccess) { err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc); delete s; return nullptr; }
    g_sets.push_back(s);
    return s;
}

struct DevMem {
    void *p = nullptr;
    int device = 0;
    hipError_t alloc(int dev, size_t bytes) { device = dev; (void)hipSetDevice(dev); return hipMalloc(&p, bytes ? bytes : 16); }
    ~DevMem() { if (p) { (void)hipSetDevice(device); (void)hipFree(p); } }
};

struct Rank {
    rrt_ctx *ctx = nullptr;
    int device = 0;
    uint32_t npix = 0;                 // pixels this rank renders (TILE: its share; SAMPLE: all)
    DevMem rgb, acc;                   // this rank's outputs on its device
    rrt_stats stats{};
    volatile uint64_t progress = 0;
    int rc = RRT_OK;
    std::string err;
};

#define RANK_TRY(expr)                                                                                 \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) { R.rc = RRT_ERR_HIP; R.err = std::string(#expr) + ": " + hipGetErrorString(e_); return; } \
    } while (0)
#define NCCL_TRY(expr)                                                                                 \
    do {                                                                                               \
        int n_ = (expr);                                                                               \
        if (n_ != kNcclSuccess) { R.rc = RRT_ERR_HIP; R.err = std::string(#expr) + ": " + g_rccl.GetErrorString(n_); return; } \
    } while (0)

}  // namespace

extern "C" int rrt_render_multi(rrt_ctx *const *ctxs, uint32_t n_ctx, const rrt_render_params *params, uint8_t *out_rgb8, float *out_accum,
                                rrt_stats *stats, volatile uint64_t *progress) {
    if (!ctxs || n_ctx == 0 || !params || !out_rgb8) return set_error(RRT_ERR_INVALID_ARGUMENT, "rrt_render_multi: null argument");
    for (uint32_t i = 0; i < n_ctx; i++) if (!ctxs[i]) return set_error(RRT_ERR_INVALID_ARGUMENT, "rrt_render_multi: null context");
    if (params->shard_mode > RRT_SHARD_SAMPLE) return set_error(RRT_ERR_INVALID_ARGUMENT, "rrt_render_multi: bad shard mode");
    const char *env_force = std::getenv("RRT_MULTI_FORCE_RCCL");
    const char *env_loop = std::getenv("RRT_MULTI_LOOPBACK");
    const bool force_rccl = env_force && env_force[0] == '1';
    // Loopback: ranks may share a device and the exchange is done with device-to-device copies instead of RCCL, which
    // refuses two ranks on one GPU.  For exercising the sharding, packing and placing on a one-GPU machine.
    const bool loopback = env_loop && env_loop[0] == '1';
    if (n_ctx == 1 && !force_rccl && !loopback) {
        rrt_render_params p = *params;
        p.shard_mode = RRT_SHARD_NONE; p.shard_rank = 0; p.shard_world = 1;
        return rrt_render(ctxs[0], &p, out_rgb8, out_accum, stats, progress);
    }
    try {
        const uint32_t n = n_ctx;
        const bool tile = params->shard_mode != RRT_SHARD_SAMPLE;   // NONE means "the default": tiles
        const uint32_t nx = params->nx, ny = params->ny;
        if (nx == 0 || ny == 0 || params->ns == 0) return set_error(RRT_ERR_INVALID_ARGUMENT, "nx, ny and ns must be positive");
        const size_t npix_img = (size_t)nx * ny;
        const bool want_acc = out_accum != nullptr;

        std::vector<Rank> ranks(n);
        std::vector<int> devices(n);
        for (uint32_t r = 0; r < n; r++) { ranks[r].ctx = ctxs[r]; ranks[r].device = devices[r] = context_device(ctxs[r]); }
        CommSet *cs = nullptr;
        if (!loopback) {
            for (uint32_t a = 0; a < n; a++)
                for (uint32_t b = a + 1; b < n; b++)
                    if (devices[a] == devices[b]) return set_error(RRT_ERR_INVALID_ARGUMENT, "rrt_render_multi: two contexts on one GPU");
            std::string err;
            cs = comm_set_for(devices, err);
            if (!cs) return set_error(RRT_ERR_HIP, "rrt_render_multi: " + err);
        }

        // ---- what each rank renders, and the buffers ----
        std::vector<std::vector<uint32_t>> maps(n);
        std::vector<size_t> offset(n + 1, 0);  // pixel offset of each rank's share inside rank 0's receive staging
        for (uint32_t r = 0; r < n; r++) {
            if (tile && n > 1) { shard_pixel_list(nx, ny, r, n, maps[r]); ranks[r].npix = (uint32_t)maps[r].size(); }
            else ranks[r].npix = (uint32_t)npix_img;
            offset[r + 1] = offset[r] + ranks[r].npix;
        }
        const bool packed = tile && n > 1;
        // (one rank, RCCL forced: its share still travels through a send and a receive, to itself)
        const bool self_exchange = tile && n == 1 && force_rccl && !loopback;
        for (uint32_t r = 0; r < n; r++) {
            Rank &R = ranks[r];
            if (tile && R.rgb.alloc(R.device, (size_t)R.npix * 3) != hipSuccess) return set_error(RRT_ERR_OUT_OF_MEMORY, "rrt_render_multi: out of device memory");
            if ((!tile || want_acc) && R.acc.alloc(R.device, (size_t)R.npix * 12) != hipSuccess) return set_error(RRT_ERR_OUT_OF_MEMORY, "rrt_render_multi: out of device memory");
        }
        // rank 0: the whole image, the peers' packed shares, and every rank's pixel list
        const int dev0 = devices[0];
        DevMem img_rgb, img_acc, stage_rgb, stage_acc, d_maps;
        if (img_rgb.alloc(dev0, npix_img * 3) != hipSuccess || ((want_acc || !tile) && img_acc.alloc(dev0, npix_img * 12) != hipSuccess))
            return set_error(RRT_ERR_OUT_OF_MEMORY, "rrt_render_multi: out of device memory");
        if (self_exchange && (stage_rgb.alloc(dev0, npix_img * 3) != hipSuccess || (want_acc && stage_acc.alloc(dev0, npix_img * 12) != hipSuccess)))
            return set_error(RRT_ERR_OUT_OF_MEMORY, "rrt_render_multi: out of device memory");
        if (packed) {
            if (stage_rgb.alloc(dev0, npix_img * 3) != hipSuccess || (want_acc && stage_acc.alloc(dev0, npix_img * 12) != hipSuccess) ||
                d_maps.alloc(dev0, npix_img * 4) != hipSuccess)
                return set_error(RRT_ERR_OUT_OF_MEMORY, "rrt_render_multi: out of device memory");
            for (uint32_t r = 0; r < n; r++)
                if (hipMemcpy(static_cast<uint32_t *>(d_maps.p) + offset[r], maps[r].data(), maps[r].size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                    return set_error(RRT_ERR_HIP, "rrt_render_multi: cannot upload the pixel lists");
        }

        // ---- one thread per rank: render, then its side of the exchange ----
        std::atomic<uint32_t> running{n};
        auto work = [&](uint32_t r) {
            Rank &R = ranks[r];
            struct Done { std::atomic<uint32_t> &c; ~Done() { c.fetch_sub(1); } } done{running};
            RANK_TRY(hipSetDevice(R.device));
            rrt_render_params p = *params;
            p.shard_mode = n == 1 ? RRT_SHARD_NONE : (tile ? RRT_SHARD_TILE : RRT_SHARD_SAMPLE);
            p.shard_rank = r; p.shard_world = n;
            R.rc = render_shard(R.ctx, &p, tile ? R.rgb.p : nullptr, R.acc.p, &R.stats, &R.progress, packed);
            if (R.rc != RRT_OK) { R.err = last_error_text(); return; }
            if (loopback || !cs) return;  // loopback: rank 0 collects below, after every rank is done
            hipStream_t st = context_stream(R.ctx);
            ncclComm_t comm = cs->comms[r];
            if (tile) {
                if (n == 1 && !self_exchange) return;
                if (self_exchange) {
                    NCCL_TRY(g_rccl.GroupStart());
                    NCCL_TRY(g_rccl.Send(R.rgb.p, (size_t)R.npix * 3, kNcclUint8, 0, comm, st));
                    NCCL_TRY(g_rccl.Recv(stage_rgb.p, (size_t)R.npix * 3, kNcclUint8, 0, comm, st));
                    if (want_acc) {
                        NCCL_TRY(g_rccl.Send(R.acc.p, (size_t)R.npix * 3, kNcclFloat32, 0, comm, st));
                        NCCL_TRY(g_rccl.Recv(stage_acc.p, (size_t)R.npix * 3, kNcclFloat32, 0, comm, st));
                    }
                    NCCL_TRY(g_rccl.GroupEnd());
                    RANK_TRY(hipStreamSynchronize(st));
                    return;
                }
                NCCL_TRY(g_rccl.GroupStart());
                if (r == 0) {
                    for (uint32_t q = 1; q < n; q++) {
                        NCCL_TRY(g_rccl.Recv(static_cast<uint8_t *>(stage_rgb.p) + offset[q] * 3, (size_t)ranks[q].npix * 3, kNcclUint8, (int)q, comm, st));
                        if (want_acc) NCCL_TRY(g_rccl.Recv(static_cast<float *>(stage_acc.p) + offset[q] * 3, (size_t)ranks[q].npix * 3, kNcclFloat32, (int)q, comm, st));
                    }
                } else {
                    NCCL_TRY(g_rccl.Send(R.rgb.p, (size_t)R.npix * 3, kNcclUint8, 0, comm, st));
                    if (want_acc) NCCL_TRY(g_rccl.Send(R.acc.p, (size_t)R.npix * 3, kNcclFloat32, 0, comm, st));
                }
                NCCL_TRY(g_rccl.GroupEnd());
            } else {
                NCCL_TRY(g_rccl.Reduce(R.acc.p, r == 0 ? img_acc.p : nullptr, npix_img * 3, kNcclFloat32, kNcclSum, 0, comm, st));
            }
            RANK_TRY(hipStreamSynchronize(st));
        };
        std::vector<std::thread> threads;
        for (uint32_t r = 0; r < n; r++) threads.emplace_back(work, r);
        while (running.load() != 0) {  // relay the ranks' progress: the sum of what each has finished (lib.rs:109)
            if (progress) {
                uint64_t sum = 0;
                for (const Rank &R : ranks) sum += R.progress;
                if (!tile) sum /= n;   // every SAMPLE rank works on all pixels
                if (sum > *progress && sum < npix_img) *progress = sum;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
        for (std::thread &t : threads) t.join();
        for (const Rank &R : ranks) if (R.rc != RRT_OK) return set_error(R.rc, "rrt_render_multi: rank failed: " + R.err);

        // ---- rank 0: put the shares in place / quantise, and hand the image to the caller ----
        if (hipSetDevice(dev0) != hipSuccess) return set_error(RRT_ERR_HIP, "hipSetDevice");
        hipStream_t s0 = context_stream(ranks[0].ctx);
        auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess) { set_error(RRT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); return false; } return true; };
        if (tile) {
            if (!packed) {
                if (!hip_ok(hipMemcpyAsync(img_rgb.p, self_exchange ? stage_rgb.p : ranks[0].rgb.p, npix_img * 3, hipMemcpyDeviceToDevice, s0), "copy")) return RRT_ERR_HIP;
                if (want_acc && !hip_ok(hipMemcpyAsync(img_acc.p, self_exchange ? stage_acc.p : ranks[0].acc.p, npix_img * 12, hipMemcpyDeviceToDevice, s0), "copy")) return RRT_ERR_HIP;
            } else {
                for (uint32_t r = 0; r < n; r++) {
                    const void *src_rgb = r == 0 ? ranks[0].rgb.p : static_cast<uint8_t *>(stage_rgb.p) + offset[r] * 3;
                    const void *src_acc = r == 0 ? ranks[0].acc.p : (want_acc ? static_cast<float *>(stage_acc.p) + offset[r] * 3 : nullptr);
                    if (loopback && r != 0) {  // the peers' shares come by plain copies, on rank 0's stream
                        if (!hip_ok(hipMemcpyAsync(static_cast<uint8_t *>(stage_rgb.p) + offset[r] * 3, ranks[r].rgb.p, (size_t)ranks[r].npix * 3, hipMemcpyDefault, s0), "copy")) return RRT_ERR_HIP;
                        if (want_acc && !hip_ok(hipMemcpyAsync(static_cast<float *>(stage_acc.p) + offset[r] * 3, ranks[r].acc.p, (size_t)ranks[r].npix * 12, hipMemcpyDefault, s0), "copy")) return RRT_ERR_HIP;
                    }
                    const uint32_t *map = static_cast<const uint32_t *>(d_maps.p) + offset[r];
                    launch_scatter_rgb8(s0, static_cast<const uint8_t *>(src_rgb), map, ranks[r].npix, static_cast<uint8_t *>(img_rgb.p));
                    if (want_acc) launch_scatter_f32(s0, static_cast<const float *>(src_acc), map, ranks[r].npix, static_cast<float *>(img_acc.p));
                }
            }
        } else {
            if (loopback || !cs) {  // sum in rank order on rank 0's device
                if (!hip_ok(hipMemcpyAsync(img_acc.p, ranks[0].acc.p, npix_img * 12, hipMemcpyDefault, s0), "copy")) return RRT_ERR_HIP;
                DevMem tmp;
                if (n > 1 && tmp.alloc(dev0, npix_img * 12) != hipSuccess) return set_error(RRT_ERR_OUT_OF_MEMORY, "rrt_render_multi: out of device memory");
                for (uint32_t r = 1; r < n; r++) {
                    if (!hip_ok(hipMemcpyAsync(tmp.p, ranks[r].acc.p, npix_img * 12, hipMemcpyDefault, s0), "copy")) return RRT_ERR_HIP;
                    launch_add_f32(s0, static_cast<float *>(img_acc.p), static_cast<const float *>(tmp.p), npix_img * 3);
                    if (!hip_ok(hipStreamSynchronize(s0), "sum")) return RRT_ERR_HIP;
                }
            }
            launch_finalize_interleaved(s0, static_cast<const float *>(img_acc.p), (uint32_t)npix_img, params->ns, static_cast<uint8_t *>(img_rgb.p));
        }
        if (!hip_ok(hipGetLastError(), "launch")) return RRT_ERR_HIP;
        if (!hip_ok(hipMemcpyAsync(out_rgb8, img_rgb.p, npix_img * 3, hipMemcpyDeviceToHost, s0), "copy")) return RRT_ERR_HIP;
        if (want_acc && !hip_ok(hipMemcpyAsync(out_accum, img_acc.p, npix_img * 12, hipMemcpyDeviceToHost, s0), "copy")) return RRT_ERR_HIP;
        if (!hip_ok(hipStreamSynchronize(s0), "sync")) return RRT_ERR_HIP;

        if (stats) {
            rrt_stats t{};
            for (const Rank &R : ranks) {
                t.rays += R.stats.rays; t.inner_visits += R.stats.inner_visits; t.leaf_tests += R.stats.leaf_tests;
                t.hits_lambertian += R.stats.hits_lambertian; t.hits_metal += R.stats.hits_metal; t.hits_dielectric += R.stats.hits_dielectric;
                t.sky_misses += R.stats.sky_misses; t.paths += R.stats.paths; t.depth_cut += R.stats.depth_cut; t.absorbed += R.stats.absorbed;
                t.iterations = R.stats.iterations > t.iterations ? R.stats.iterations : t.iterations;
                t.gpu_ms = R.stats.gpu_ms > t.gpu_ms ? R.stats.gpu_ms : t.gpu_ms;
                t.extend_ms += R.stats.extend_ms; t.extend_launches += R.stats.extend_launches;
            }
            *stats = t;
        }
        if (progress) *progress = npix_img;
        return RRT_OK;
    } catch (const std::exception &e) {
        return set_error(RRT_ERR_OUT_OF_MEMORY, std::string("rrt_render_multi: ") + e.what());
    }
}
