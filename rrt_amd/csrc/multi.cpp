// rrt_render_multi: one call, whole image back (lib.rs:73-81), rendered on several GPUs of one node.
//
// One host thread per context renders that rank's shard (TILE: 32x8 tiles dealt round-robin; SAMPLE: sample k -> rank
// k mod n); once every rank is back — and only if none of them failed — the framebuffers are combined on rank 0's GPU
// over RCCL, all ranks' calls in one group issued by the calling thread:
//   TILE    every peer sends the RGB8 (and, if asked for, f32) pixels it owns, packed, straight to rank 0 — one ncclSend
//           per peer and plane inside one group, 24.9 MB of RGB8 in all at 3840x2160; rank 0 puts them in place;
//   SAMPLE  ncclReduce(sum) of the f32 sums onto rank 0, which quantises once (lib.rs:105-108).
// xGMI links every pair of GPUs directly, so one hop per peer is the whole exchange.  The work counters are summed on the
// host: all ranks live in this process.
//
// RCCL is loaded on first use (dlopen), so programs that render on one GPU never load it.  Worker threads, staging buffers
// and the ranks' pixel lists are kept per device list from call to call (Plan), like the communicators.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdio>
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

// Device memory that grows on demand and is kept from call to call.
struct DevMem {
    void *p = nullptr;
    size_t cap = 0;
    int device = 0;
    hipError_t ensure(int dev, size_t bytes) {
        if (p && dev == device && bytes <= cap) return hipSuccess;
        release();
        device = dev;
        (void)hipSetDevice(dev);
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e == hipSuccess) cap = bytes ? bytes : 16; else p = nullptr;
        return e;
    }
    void release() { if (p) { (void)hipSetDevice(device); (void)hipFree(p); } p = nullptr; cap = 0; }
};

// One host thread per rank, kept between calls: it takes one job at a time.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false;
    Worker() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lock(m);
            for (;;) {
                cv.wait(lock, [this] { return (bool)job; });
                std::function<void()> f = std::move(job);
                job = nullptr;
                lock.unlock();
                f();
                lock.lock();
                busy = false;
                cv.notify_all();
            }
        });
        th.detach();  // lives as long as the process: plans are never torn down
    }
    void start(std::function<void()> f) {
        std::lock_guard<std::mutex> lock(m);
        busy = true;
        job = std::move(f);
        cv.notify_all();
    }
    bool running() { std::lock_guard<std::mutex> lock(m); return busy; }
    void wait() { std::unique_lock<std::mutex> lock(m); cv.wait(lock, [this] { return !busy; }); }
};

struct Rank {
    rrt_ctx *ctx = nullptr;
    int device = 0;
    uint32_t npix = 0;                 // pixels this rank renders (TILE: its share; SAMPLE: all)
    DevMem rgb, acc;                   // this rank's outputs on its device
    rrt_stats stats{};
    uint64_t progress = 0;             // written by the rank's thread (plain stores of the render loop), read by the caller's
    int rc = RRT_OK;
    std::string err;
};

// What a set of ranks needs from call to call, per device list (and loopback or not): the worker threads, every buffer,
// the ranks' pixel lists on rank 0's GPU.  Made once, resized when the image or the sharding changes, never torn down.
struct Plan {
    std::vector<int> devices;
    bool loopback = false;
    std::mutex busy;                          // one rrt_render_multi at a time per plan
    std::vector<Rank> ranks;
    std::vector<Worker *> workers;
    DevMem img_rgb, img_acc, stage_rgb, stage_acc, d_maps, tmp;   // on rank 0's device
    std::vector<size_t> offset;               // pixel offset of each rank's share inside rank 0's staging
    uint32_t maps_nx = 0, maps_ny = 0;        // the image the pixel lists in d_maps were made for (0: none)
};
std::vector<Plan *> g_plans;

Plan *plan_for(const std::vector<int> &devices, bool loopback) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (Plan *p : g_plans) if (p->devices == devices && p->loopback == loopback) return p;
    Plan *p = new Plan();
    p->devices = devices;
    p->loopback = loopback;
    p->ranks.resize(devices.size());
    for (size_t r = 0; r < devices.size(); r++) p->workers.push_back(new Worker());
    p->offset.assign(devices.size() + 1, 0);
    g_plans.push_back(p);
    return p;
}

uint64_t peek_u64(const uint64_t *p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
void poke_u64(volatile uint64_t *p, uint64_t v) { __atomic_store_n(const_cast<uint64_t *>(p), v, __ATOMIC_RELAXED); }

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
        // RRT_MULTI_TIMING=1: the call's phases on stderr (set-up, rendering, exchange, placing + copy to the host), in ms
        const char *env_timing = std::getenv("RRT_MULTI_TIMING");
        const bool timing = env_timing && env_timing[0] == '1';
        const auto t_begin = std::chrono::steady_clock::now();
        auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
        const bool tile = params->shard_mode != RRT_SHARD_SAMPLE;   // NONE means "the default": tiles
        const uint32_t nx = params->nx, ny = params->ny;
        if (nx == 0 || ny == 0 || params->ns == 0) return set_error(RRT_ERR_INVALID_ARGUMENT, "nx, ny and ns must be positive");
        if ((uint64_t)nx * ny > (1ull << 28)) return set_error(RRT_ERR_INVALID_ARGUMENT, "image too large");
        const size_t npix_img = (size_t)nx * ny;
        const bool want_acc = out_accum != nullptr;
        if (progress) poke_u64(progress, 0);

        std::vector<int> devices(n);
        for (uint32_t r = 0; r < n; r++) devices[r] = context_device(ctxs[r]);
        CommSet *cs = nullptr;
        if (!loopback) {
            for (uint32_t a = 0; a < n; a++)
                for (uint32_t b = a + 1; b < n; b++)
                    if (devices[a] == devices[b]) return set_error(RRT_ERR_INVALID_ARGUMENT, "rrt_render_multi: two contexts on one GPU");
            std::string err;
            cs = comm_set_for(devices, err);
            if (!cs) return set_error(RRT_ERR_HIP, "rrt_render_multi: " + err);
        }
        Plan &pl = *plan_for(devices, loopback);
        std::lock_guard<std::mutex> plan_lock(pl.busy);
        std::vector<Rank> &ranks = pl.ranks;
        std::vector<size_t> &offset = pl.offset;
        const int dev0 = devices[0];
        for (uint32_t r = 0; r < n; r++) { ranks[r].ctx = ctxs[r]; ranks[r].device = devices[r]; ranks[r].rc = RRT_OK; ranks[r].err.clear(); ranks[r].progress = 0; ranks[r].stats = rrt_stats{}; }

        // ---- what each rank renders, and the buffers (kept from the last call wherever they still fit) ----
        const bool packed = tile && n > 1;
        // (one rank, RCCL forced: its share still travels through a send and a receive, to itself)
        const bool self_exchange = tile && n == 1 && force_rccl && !loopback;
        auto oom = [] { return set_error(RRT_ERR_OUT_OF_MEMORY, "rrt_render_multi: out of device memory"); };
        if (packed && !(pl.maps_nx == nx && pl.maps_ny == ny)) {  // the ranks' pixel lists, side by side on rank 0's GPU
            pl.maps_nx = pl.maps_ny = 0;
            if (pl.d_maps.ensure(dev0, npix_img * 4) != hipSuccess) return oom();
            std::vector<uint32_t> map;
            for (uint32_t r = 0; r < n; r++) {
                shard_pixel_list(nx, ny, r, n, map);
                ranks[r].npix = (uint32_t)map.size();
                offset[r + 1] = offset[r] + map.size();
                if (hipMemcpy(static_cast<uint32_t *>(pl.d_maps.p) + offset[r], map.data(), map.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                    return set_error(RRT_ERR_HIP, "rrt_render_multi: cannot upload the pixel lists");
            }
            pl.maps_nx = nx; pl.maps_ny = ny;
        }
        if (!packed) {
            pl.maps_nx = pl.maps_ny = 0;
            for (uint32_t r = 0; r < n; r++) { ranks[r].npix = (uint32_t)npix_img; offset[r + 1] = offset[r] + npix_img; }
        }
        for (uint32_t r = 0; r < n; r++) {
            Rank &R = ranks[r];
            if (tile && R.rgb.ensure(R.device, (size_t)R.npix * 3) != hipSuccess) return oom();
            if ((!tile || want_acc) && R.acc.ensure(R.device, (size_t)R.npix * 12) != hipSuccess) return oom();
        }
        // rank 0: the whole image and the peers' packed shares
        if (pl.img_rgb.ensure(dev0, npix_img * 3) != hipSuccess || ((want_acc || !tile) && pl.img_acc.ensure(dev0, npix_img * 12) != hipSuccess)) return oom();
        if ((self_exchange || packed) && (pl.stage_rgb.ensure(dev0, npix_img * 3) != hipSuccess || (want_acc && pl.stage_acc.ensure(dev0, npix_img * 12) != hipSuccess)))
            return oom();
        if (!tile && n > 1 && (loopback || !cs) && pl.tmp.ensure(dev0, npix_img * 12) != hipSuccess) return oom();

        const double ms_setup = ms_since(t_begin);
        const auto t_render = std::chrono::steady_clock::now();
        // ---- every rank renders its shard, on its own thread ----
        for (uint32_t r = 0; r < n; r++) {
            pl.workers[r]->start([&, r] {
                Rank &R = ranks[r];
                if (hipSetDevice(R.device) != hipSuccess) { R.rc = RRT_ERR_HIP; R.err = "hipSetDevice"; return; }
                rrt_render_params p = *params;
                p.shard_mode = n == 1 ? RRT_SHARD_NONE : (tile ? RRT_SHARD_TILE : RRT_SHARD_SAMPLE);
                p.shard_rank = r; p.shard_world = n;
                R.rc = render_shard(R.ctx, &p, tile ? R.rgb.p : nullptr, R.acc.p, &R.stats, &R.progress, packed);
                if (R.rc != RRT_OK) R.err = last_error_text();
            });
        }
        for (;;) {  // relay the ranks' progress: the sum of what each has finished (lib.rs:109)
            bool any = false;
            for (Worker *w : pl.workers) any = any || w->running();
            if (!any) break;
            if (progress) {
                uint64_t sum = 0;
                for (const Rank &R : ranks) sum += peek_u64(&R.progress);
                if (!tile) sum /= n;   // every SAMPLE rank works on all pixels
                if (sum > peek_u64(const_cast<const uint64_t *>(progress)) && sum < npix_img) poke_u64(progress, sum);
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        for (Worker *w : pl.workers) w->wait();
        // Every rank is back before any of them enters the exchange: a rank that failed would leave its peers waiting in a
        // receive or a reduction that is never matched.
        for (const Rank &R : ranks) if (R.rc != RRT_OK) return set_error(R.rc, "rrt_render_multi: rank failed: " + R.err);

        const double ms_render = ms_since(t_render);
        const auto t_exchange = std::chrono::steady_clock::now();
        // ---- the exchange: all ranks' calls in one group, made from this thread (every rank's stream is idle) ----
        if (!loopback && cs && !(tile && n == 1 && !self_exchange)) {
            int nerr = kNcclSuccess;
            const char *where = "";
            auto note = [&](int e, const char *what) { if (e != kNcclSuccess && nerr == kNcclSuccess) { nerr = e; where = what; } };
            note(g_rccl.GroupStart(), "ncclGroupStart");
            if (nerr == kNcclSuccess) {
                for (uint32_t r = 0; r < n; r++) {
                    Rank &R = ranks[r];
                    (void)hipSetDevice(R.device);
                    hipStream_t st = context_stream(R.ctx);
                    ncclComm_t comm = cs->comms[r];
                    if (self_exchange) {
                        note(g_rccl.Send(R.rgb.p, (size_t)R.npix * 3, kNcclUint8, 0, comm, st), "ncclSend");
                        note(g_rccl.Recv(pl.stage_rgb.p, (size_t)R.npix * 3, kNcclUint8, 0, comm, st), "ncclRecv");
                        if (want_acc) {
                            note(g_rccl.Send(R.acc.p, (size_t)R.npix * 3, kNcclFloat32, 0, comm, st), "ncclSend");
                            note(g_rccl.Recv(pl.stage_acc.p, (size_t)R.npix * 3, kNcclFloat32, 0, comm, st), "ncclRecv");
                        }
                    } else if (tile) {
                        if (r == 0) {
                            for (uint32_t q = 1; q < n; q++) {
                                note(g_rccl.Recv(static_cast<uint8_t *>(pl.stage_rgb.p) + offset[q] * 3, (size_t)ranks[q].npix * 3, kNcclUint8, (int)q, comm, st), "ncclRecv");
                                if (want_acc) note(g_rccl.Recv(static_cast<float *>(pl.stage_acc.p) + offset[q] * 3, (size_t)ranks[q].npix * 3, kNcclFloat32, (int)q, comm, st), "ncclRecv");
                            }
                        } else {
                            note(g_rccl.Send(R.rgb.p, (size_t)R.npix * 3, kNcclUint8, 0, comm, st), "ncclSend");
                            if (want_acc) note(g_rccl.Send(R.acc.p, (size_t)R.npix * 3, kNcclFloat32, 0, comm, st), "ncclSend");
                        }
                    } else {
                        note(g_rccl.Reduce(R.acc.p, r == 0 ? pl.img_acc.p : nullptr, npix_img * 3, kNcclFloat32, kNcclSum, 0, comm, st), "ncclReduce");
                    }
                }
                note(g_rccl.GroupEnd(), "ncclGroupEnd");  // also after a failed call: the group must not stay open
            }
            hipError_t herr = hipSuccess;
            for (uint32_t r = 0; r < n; r++) {
                (void)hipSetDevice(ranks[r].device);
                const hipError_t e = hipStreamSynchronize(context_stream(ranks[r].ctx));
                if (e != hipSuccess && herr == hipSuccess) herr = e;
            }
            if (nerr != kNcclSuccess) return set_error(RRT_ERR_HIP, std::string("rrt_render_multi: ") + where + ": " + g_rccl.GetErrorString(nerr));
            if (herr != hipSuccess) return set_error(RRT_ERR_HIP, std::string("rrt_render_multi: exchange: ") + hipGetErrorString(herr));
        }

        const double ms_exchange = ms_since(t_exchange);
        const auto t_place = std::chrono::steady_clock::now();
        // ---- rank 0: put the shares in place / quantise, and hand the image to the caller ----
        if (hipSetDevice(dev0) != hipSuccess) return set_error(RRT_ERR_HIP, "hipSetDevice");
        hipStream_t s0 = context_stream(ranks[0].ctx);
        auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess) { set_error(RRT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); return false; } return true; };
        if (tile) {
            if (!packed) {
                if (!hip_ok(hipMemcpyAsync(pl.img_rgb.p, self_exchange ? pl.stage_rgb.p : ranks[0].rgb.p, npix_img * 3, hipMemcpyDeviceToDevice, s0), "copy")) return RRT_ERR_HIP;
                if (want_acc && !hip_ok(hipMemcpyAsync(pl.img_acc.p, self_exchange ? pl.stage_acc.p : ranks[0].acc.p, npix_img * 12, hipMemcpyDeviceToDevice, s0), "copy")) return RRT_ERR_HIP;
            } else {
                for (uint32_t r = 0; r < n; r++) {
                    const void *src_rgb = r == 0 ? ranks[0].rgb.p : static_cast<uint8_t *>(pl.stage_rgb.p) + offset[r] * 3;
                    const void *src_acc = r == 0 ? ranks[0].acc.p : (want_acc ? static_cast<float *>(pl.stage_acc.p) + offset[r] * 3 : nullptr);
                    if (loopback && r != 0) {  // the peers' shares come by plain copies, on rank 0's stream
                        if (!hip_ok(hipMemcpyAsync(static_cast<uint8_t *>(pl.stage_rgb.p) + offset[r] * 3, ranks[r].rgb.p, (size_t)ranks[r].npix * 3, hipMemcpyDefault, s0), "copy")) return RRT_ERR_HIP;
                        if (want_acc && !hip_ok(hipMemcpyAsync(static_cast<float *>(pl.stage_acc.p) + offset[r] * 3, ranks[r].acc.p, (size_t)ranks[r].npix * 12, hipMemcpyDefault, s0), "copy")) return RRT_ERR_HIP;
                    }
                    const uint32_t *map = static_cast<const uint32_t *>(pl.d_maps.p) + offset[r];
                    launch_scatter_rgb8(s0, static_cast<const uint8_t *>(src_rgb), map, ranks[r].npix, static_cast<uint8_t *>(pl.img_rgb.p));
                    if (want_acc) launch_scatter_f32(s0, static_cast<const float *>(src_acc), map, ranks[r].npix, static_cast<float *>(pl.img_acc.p));
                }
            }
        } else {
            if (loopback || !cs) {  // sum in rank order on rank 0's device
                if (!hip_ok(hipMemcpyAsync(pl.img_acc.p, ranks[0].acc.p, npix_img * 12, hipMemcpyDefault, s0), "copy")) return RRT_ERR_HIP;
                for (uint32_t r = 1; r < n; r++) {
                    if (!hip_ok(hipMemcpyAsync(pl.tmp.p, ranks[r].acc.p, npix_img * 12, hipMemcpyDefault, s0), "copy")) return RRT_ERR_HIP;
                    launch_add_f32(s0, static_cast<float *>(pl.img_acc.p), static_cast<const float *>(pl.tmp.p), npix_img * 3);
                }
            }
            launch_finalize_interleaved(s0, static_cast<const float *>(pl.img_acc.p), (uint32_t)npix_img, params->ns, static_cast<uint8_t *>(pl.img_rgb.p));
        }
        if (!hip_ok(hipGetLastError(), "launch")) return RRT_ERR_HIP;
        if (!hip_ok(hipMemcpyAsync(out_rgb8, pl.img_rgb.p, npix_img * 3, hipMemcpyDeviceToHost, s0), "copy")) return RRT_ERR_HIP;
        if (want_acc && !hip_ok(hipMemcpyAsync(out_accum, pl.img_acc.p, npix_img * 12, hipMemcpyDeviceToHost, s0), "copy")) return RRT_ERR_HIP;
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
        if (progress) poke_u64(progress, npix_img);
        if (timing)
            std::fprintf(stderr, "rrt_render_multi: set-up %.3f ms, rendering %.3f ms, exchange %.3f ms, placing and copy to the host %.3f ms\n", ms_setup, ms_render,
                         ms_exchange, ms_since(t_place));
        return RRT_OK;
    } catch (const std::exception &e) {
        return set_error(RRT_ERR_OUT_OF_MEMORY, std::string("rrt_render_multi: ") + e.what());
    }
}
