// rrt_render_device_noalloc: rrt_render_device for callers that cannot allocate device memory (code under HIP graph
// capture, tight serving loops, harnesses that forbid hipMalloc).  Everything the render needs beyond the caller's
// output buffers — the path pool, its queues and counters, the radiance planes, the per-pixel sums — is carved out of
// one array that is part of the library's code object, so every GPU the library runs on holds 16 MiB of it from the
// moment the code object is loaded there until the process ends.
//
// `scratch` below is a host-side stand-in for that array, not a device address: each device has its own copy at its own
// address, which hipGetSymbolAddress returns with that device current.  Host code uses those addresses and never the
// array's name.
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>

#include "../../include/rrt_amd.h"

namespace rrt_amd {
int render_in_workspace(rrt_ctx *ctx, const rrt_render_params *params, void *d_rgb8, void *d_accum, rrt_stats *stats, void *workspace, size_t workspace_bytes);
int context_device(const rrt_ctx *ctx);
int set_error(int code, const std::string &msg);
}  // namespace rrt_amd

namespace {

constexpr size_t kScratchBytes = size_t(RRT_NOALLOC_WORKSPACE_BYTES);
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// (The kernels of the render path take the workspace as pointers in their arguments and nothing refers to the array by
// name in device code.  The HIP_SYMBOL use in static_workspace() below is what makes the compiler keep the array in the
// code object and give it a name the loader can look up.)

constexpr int kMaxDevices = 64;
std::mutex g_mutex;
void *g_ptr[kMaxDevices] = {};  // per device: address of its copy of `scratch`

}  // namespace

namespace rrt_amd {

// Called by rrt_create with `device` current: looks the device's copy up once and remembers it.
int static_workspace(int device, void **ptr, size_t *bytes, std::string &err) {
    if (device < 0 || device >= kMaxDevices) { err = "device index beyond the static workspace table"; return RRT_ERR_INVALID_ARGUMENT; }
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_ptr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) { err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e); return RRT_ERR_HIP; }
        g_ptr[device] = p;
    }
    if (ptr) *ptr = g_ptr[device];
    if (bytes) *bytes = kScratchBytes;
    return RRT_OK;
}

}  // namespace rrt_amd

extern "C" int rrt_render_device_noalloc(rrt_ctx *ctx, const rrt_render_params *params, void *d_rgb8, void *d_accum, rrt_stats *stats) {
    const int device = rrt_amd::context_device(ctx);
    if (device < 0 || device >= kMaxDevices) return rrt_amd::set_error(RRT_ERR_INVALID_ARGUMENT, "rrt_render_device_noalloc: bad context");
    void *ws = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        ws = g_ptr[device];  // set by rrt_create on this device
    }
    if (!ws) return rrt_amd::set_error(RRT_ERR_NOT_READY, "rrt_render_device_noalloc: no static workspace on this device");
    return rrt_amd::render_in_workspace(ctx, params, d_rgb8, d_accum, stats, ws, kScratchBytes);
}
