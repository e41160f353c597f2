// Does an LDS-direct load (global_load_lds_dwordx4) gather from a per-lane source address?  Lane i of the wave lands in
// entry base + i of the LDS array, whatever address it read from.  hipcc -O3 --offload-arch=gfx950 glds_gather.hip -o glds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const uint4 *src, const uint32_t *idx, uint4 *out) {
    __shared__ uint4 ent[4][256];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (uint32_t t = 0; t < 4; t++) {
        const uint32_t i = idx[w * 256 + t * 64 + lane];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i),
                                         (__attribute__((address_space(3))) void *)&ent[w][t * 64], 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);  // everything
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t t = 0; t < 4; t++) out[w * 256 + t * 64 + lane] = ent[w][t * 64 + lane];
}
int main() {
    const int n = 1024;
    uint4 *src, *out; uint32_t *idx;
    hipMallocManaged(&src, n * 16); hipMallocManaged(&out, n * 16); hipMallocManaged(&idx, n * 4);
    for (int i = 0; i < n; i++) { src[i] = make_uint4(i, i * 3, i * 7, ~i); idx[i] = (i * 37u + 11u) % n; out[i] = make_uint4(0,0,0,0); }
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, src, idx, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("fail\n"); return 1; }
    int bad = 0;
    for (int i = 0; i < n; i++) { uint32_t j = idx[i]; if (out[i].x != j || out[i].y != j * 3 || out[i].z != j * 7 || out[i].w != ~j) bad++; }
    printf("bad %d of %d\n", bad, n);
    return bad != 0;
}
