// What a 64-lane gather of 16 bytes per lane costs the texture-address / L1 path of a CU, by how the lanes' addresses fall:
//   mode 0  every lane reads its own random 64-byte record (one 16-byte piece of it)      -> 64 lines per instruction
//   mode 1  four adjacent lanes read the four pieces of ONE random record                  -> 16 lines per instruction
//   mode 2  every lane reads all four pieces of its own record (4 instructions)            -> 64 lines, 4 instructions
//   mode 3  consecutive lanes read consecutive pieces (a plain streaming load)             -> 8 lines per instruction
// The table (2 MB by default) stays in L2, so the numbers are address-path cost, not DRAM.  Prints cycles per wave
// instruction per CU (all SIMDs of a CU share one address path).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/gather_rate.hip -o tools/micro/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(1024) void k(const uint4 *tab, uint32_t n_rec, uint32_t iters, int mode, uint4 *out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        uint32_t rec = (x >> 8) % n_rec;
        if (mode == 1) rec = __shfl((int)rec, (int)(lane & ~3u), 64);  // the quad's first lane decides
        if (mode == 3) rec = ((x >> 8) % (n_rec / 16u)) * 16u;         // wave-uniform start would need a broadcast; per-lane start, consecutive pieces below
        uint4 v;
        if (mode == 0) v = tab[rec * 4u + (x & 3u)];
        else if (mode == 1) v = tab[rec * 4u + (lane & 3u)];
        else if (mode == 2) {
            const uint4 a = tab[rec * 4u], b = tab[rec * 4u + 1u], c = tab[rec * 4u + 2u], d = tab[rec * 4u + 3u];
            v = make_uint4(a.x ^ b.x ^ c.x ^ d.x, a.y ^ b.y, c.z ^ d.z, a.w ^ d.w);
        } else {
            const uint32_t start = __shfl((int)rec, 0, 64);
            v = tab[(start * 4u + lane) % (n_rec * 4u)];
        }
        acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
        x ^= v.x & 1u;  // the next address depends on the data: one gather in flight per lane, as in a tree walk
    }
    out[blockIdx.x * 1024u + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const uint32_t mb = argc > 1 ? (uint32_t)atoi(argv[1]) : 2u;
    const uint32_t n_rec = mb * 1024u * 1024u / 64u;
    std::vector<uint4> h(n_rec * 4u);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_uint4((uint32_t)i * 2654435761u, (uint32_t)i, 7u, 1u);
    uint4 *tab, *out;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount, blocks = cus * 2;
    if (hipMalloc(&tab, h.size() * 16) != hipSuccess || hipMalloc(&out, (size_t)blocks * 1024 * 16) != hipSuccess) return 1;
    if (hipMemcpy(tab, h.data(), h.size() * 16, hipMemcpyHostToDevice) != hipSuccess) return 1;
    const uint32_t iters = 4000;
    printf("table %u MB, %d CUs, %d workgroups of 1024 (8 waves per SIMD), %u dependent gathers per lane, clock %d MHz\n", mb, cus, blocks, iters, prop.clockRate / 1000);
    for (int mode = 0; mode < 4; mode++) {
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
        k<<<blocks, 1024>>>(tab, n_rec, 200, mode, out);
        (void)hipEventRecord(e0);
        k<<<blocks, 1024>>>(tab, n_rec, iters, mode, out);
        (void)hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double waves_per_cu = 32.0, instr = waves_per_cu * iters * (mode == 2 ? 4 : 1);
        const double cycles = ms * 1e-3 * prop.clockRate * 1e3;
        printf("mode %d: %.3f ms, %.1f cycles per wave instruction per CU, %.1f cycles per 64 records\n", mode, ms, cycles / instr,
               cycles / (waves_per_cu * iters) * (mode == 1 ? 4.0 : 1.0));
    }
    return 0;
}
