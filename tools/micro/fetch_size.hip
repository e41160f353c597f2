// What rocprofv3's FETCH_SIZE and WRITE_SIZE report for coalesced streams of 4, 8, 12 and 16 bytes per lane (the widths the
// render kernels use: list entries and depth words; hit records and the narrow trail window; the packed origin; direction,
// generator state and radiance), and for 16-byte loads that go straight to LDS.  Every kernel moves the same 1 GiB (four
// times the Infinity Cache), so bytes / (counter x 1024) is the factor to apply to that width.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_size.hip -o tools/micro/fetch_size
//   rocprofv3 --pmc FETCH_SIZE -- tools/micro/fetch_size      (and again with --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int W> struct Rec { uint32_t w[W / 4]; };

template <int W>
__global__ __launch_bounds__(256) void k_read(const Rec<W> *p, size_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const Rec<W> r = p[i];
#pragma unroll
        for (int k = 0; k < W / 4; k++) acc ^= r.w[k];
    }
    if (acc == 0x12345678u) out[0] = acc;  // keeps the loads
}

template <int W>
__global__ __launch_bounds__(256) void k_write(Rec<W> *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        Rec<W> r;
#pragma unroll
        for (int k = 0; k < W / 4; k++) r.w[k] = (uint32_t)i + k;
        p[i] = r;
    }
}

// 16 bytes per lane from memory into LDS without passing through registers (as the shade kernel fetches generator states)
__global__ __launch_bounds__(256) void k_read_lds16(const uint4 *p, size_t n, uint32_t *out) {
    __shared__ uint4 buf[256];
    uint32_t acc = 0;
    for (size_t i0 = (size_t)blockIdx.x * 256u; i0 < n; i0 += (size_t)gridDim.x * 256u) {
        const size_t i = i0 + threadIdx.x;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p + (i < n ? i : 0)),
                                         (__attribute__((address_space(3))) void *)(buf + (threadIdx.x & ~63u)), 16, 0, 0);
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= buf[threadIdx.x].x;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void *buf; uint32_t *out;
    if (hipMalloc(&buf, bytes + 64) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    if (hipMemset(buf, 1, bytes) != hipSuccess) return 1;
    const dim3 grid(256 * 16), blk(256);
    hipLaunchKernelGGL(k_read<4>, grid, blk, 0, 0, (const Rec<4> *)buf, bytes / 4, out);
    hipLaunchKernelGGL(k_read<8>, grid, blk, 0, 0, (const Rec<8> *)buf, bytes / 8, out);
    hipLaunchKernelGGL(k_read<12>, grid, blk, 0, 0, (const Rec<12> *)buf, bytes / 12, out);
    hipLaunchKernelGGL(k_read<16>, grid, blk, 0, 0, (const Rec<16> *)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_read_lds16, grid, blk, 0, 0, (const uint4 *)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_write<4>, grid, blk, 0, 0, (Rec<4> *)buf, bytes / 4);
    hipLaunchKernelGGL(k_write<8>, grid, blk, 0, 0, (Rec<8> *)buf, bytes / 8);
    hipLaunchKernelGGL(k_write<12>, grid, blk, 0, 0, (Rec<12> *)buf, bytes / 12);
    hipLaunchKernelGGL(k_write<16>, grid, blk, 0, 0, (Rec<16> *)buf, bytes / 16);
    if (hipDeviceSynchronize() != hipSuccess) { printf("a kernel failed\n"); return 1; }
    printf("each kernel moved %zu bytes\n", bytes);
    return 0;
}
