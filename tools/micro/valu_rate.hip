// Micro-benchmark: issue cost of wave64 VALU / LDS instructions on gfx950 at a given occupancy.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
// Each kernel runs ITER iterations of 32 independent instructions of one kind per wave, on every SIMD of the chip at
// 1..8 waves per SIMD.  Reported: cycles per wave-instruction per SIMD = wall time x clock / instructions per SIMD, with
// the clock the kernel itself measures (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 8192;
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
// four independent destinations per line, eight lines: 32 instructions
#define F4(OP, TAIL) { REP8(asm volatile(OP " %0, %0" TAIL "\n " OP " %1, %1" TAIL "\n " OP " %2, %2" TAIL "\n " OP " %3, %3" TAIL "\n" \
                                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(q.x), "v"(q.y));) }
#define U4(OP, TAIL) { REP8(asm volatile(OP " %0, %0" TAIL "\n " OP " %1, %1" TAIL "\n " OP " %2, %2" TAIL "\n " OP " %3, %3" TAIL "\n" \
                                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u4), "v"(u5));) }
#define P4(OP, TAIL) { REP8(asm volatile(OP " %0, %0" TAIL "\n " OP " %1, %1" TAIL "\n " OP " %2, %2" TAIL "\n " OP " %3, %3" TAIL "\n" \
                                       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));) }

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *clk) {
    __shared__ float lds[4096];  // 16 KB: eight workgroups fit a CU
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    float a0 = threadIdx.x * 0.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a0, a3}, p3 = {a1, a2}, q = {1.0001f, 0.9999f};
    unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 & 7, u5 = u0 | 1;
    // LDS addresses: (a) consecutive 8-byte words, (b) pseudo-random 56-byte (or 64-byte) nodes among 256
    unsigned la = (threadIdx.x & 63) * 8, lr = ((threadIdx.x * 2654435761u) >> 24) * 56;
    f2 l0, l1, l2, l3;
    float4 w0, w1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITER; i++) {
        if (KIND == 0) F4("v_fma_f32", ", %4, %5")
        else if (KIND == 1) F4("v_add_f32", ", %4")
        else if (KIND == 2) F4("v_mul_f32", ", %4")
        else if (KIND == 3) F4("v_sub_f32", ", %4")
        else if (KIND == 4) F4("v_max_f32", ", %4")
        else if (KIND == 5) F4("v_min_f32", ", %4")
        else if (KIND == 6) F4("v_max3_f32", ", %4, %5")
        else if (KIND == 7) F4("v_med3_f32", ", %4, %5")
        else if (KIND == 8) P4("v_pk_mul_f32", ", %4")
        else if (KIND == 9) P4("v_pk_add_f32", ", %4")
        else if (KIND == 10) P4("v_pk_fma_f32", ", %4, %4")
        else if (KIND == 11) U4("v_add_u32", ", %4")
        else if (KIND == 12) U4("v_and_b32", ", %4")
        else if (KIND == 13) U4("v_or_b32", ", %4")
        else if (KIND == 14) U4("v_xor_b32", ", %4")
        else if (KIND == 15) U4("v_lshlrev_b32", ", %4")   // dst = src1 << src0: here dst = u4 << dst... operand order (shift, value)
        else if (KIND == 16) U4("v_lshl_add_u32", ", 3, %4")
        else if (KIND == 17) U4("v_bfe_u32", ", %4, 14")
        else if (KIND == 18) U4("v_add3_u32", ", %4, %5")
        else if (KIND == 19) U4("v_lshl_or_b32", ", 3, %4")
        else if (KIND == 20) U4("v_and_or_b32", ", %4, %5")
        else if (KIND == 21) U4("v_mad_u32_u24", ", %4, %5")
        else if (KIND == 22) U4("v_alignbit_b32", ", %4, 21")
        else if (KIND == 23) U4("v_perm_b32", ", %4, %5")
        else if (KIND == 24) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        else if (KIND == 25) { REP8(asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cmp_le_f32 vcc, %1, %2\n v_cmp_le_f32 vcc, %2, %3\n v_cmp_le_f32 vcc, %3, %0\n" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        else if (KIND == 26) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : );) }
        else if (KIND == 27) { REP8(asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n v_cmp_lt_u32 s[22:23], %1, %2\n v_cmp_lt_u32 s[24:25], %2, %3\n v_cmp_lt_u32 s[26:27], %3, %0\n" : : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        else if (KIND == 28) F4("v_rcp_f32", "")
        else if (KIND == 29) F4("v_sqrt_f32", "")
        else if (KIND == 30) F4("v_cvt_f32_u32", "")
        else if (KIND == 31) F4("v_div_fixup_f32", ", %4, %5")
        else if (KIND == 32) { REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)\n" : "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3) : "v"(la));) }
        else if (KIND == 33) { REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:12\n ds_read_b64 %2, %4 offset:24\n ds_read_b32 %3, %4 offset:36\n s_waitcnt lgkmcnt(0)\n" : "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3.x) : "v"(lr));) lr = (lr * 5u + 56u * 7u) % (256u * 56u); }
        else if (KIND == 34) { REP8(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %2 offset:32\n ds_read_b128 %1, %2 offset:48\n s_waitcnt lgkmcnt(0)\n" : "=v"(w0), "=v"(w1) : "v"((lr / 56u) * 64u));) lr = (lr * 5u + 56u * 7u) % (256u * 56u); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y + u0 + u1 + u2 + u3 + l0.x + l1.x + l2.x + l3.x + w0.x + w1.x;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int KIND>
void run(const char *name, float *out, unsigned long long *clk) {
    printf("%-18s", name);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = 256 * w;  // 256 CUs x (4 waves per block = 1 per SIMD) x w
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k<KIND><<<blocks, 256>>>(out, clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<KIND><<<blocks, 256>>>(out, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * 2);
        CHECK(hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0;
        for (int b = 0; b < blocks; b++) { cyc += h[b * 2]; rt += h[b * 2 + 1]; }
        const double ghz = cyc / (rt * 10.0);
        const double per_simd = (double)ITER * 32 * w;
        printf("  w%d %5.2f (%.2f GHz)", w, (ms - 0.006) * 1e-3 * ghz * 1e9 / per_simd, ghz);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    printf("\n");
}

int main() {
    float *out; unsigned long long *clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4)); CHECK(hipMalloc(&clk, 256 * 8 * 16));
    printf("cycles per wave64 instruction per SIMD at w waves per SIMD (wall time x measured clock / instructions)\n");
    run<0>("v_fma_f32", out, clk); run<1>("v_add_f32", out, clk); run<2>("v_mul_f32", out, clk); run<3>("v_sub_f32", out, clk);
    run<4>("v_max_f32", out, clk); run<5>("v_min_f32", out, clk); run<6>("v_max3_f32", out, clk); run<7>("v_med3_f32", out, clk);
    run<8>("v_pk_mul_f32", out, clk); run<9>("v_pk_add_f32", out, clk); run<10>("v_pk_fma_f32", out, clk);
    run<11>("v_add_u32", out, clk); run<12>("v_and_b32", out, clk); run<13>("v_or_b32", out, clk); run<14>("v_xor_b32", out, clk);
    run<15>("v_lshlrev_b32", out, clk); run<16>("v_lshl_add_u32", out, clk); run<17>("v_bfe_u32", out, clk); run<18>("v_add3_u32", out, clk);
    run<19>("v_lshl_or_b32", out, clk); run<20>("v_and_or_b32", out, clk); run<21>("v_mad_u32_u24", out, clk); run<22>("v_alignbit_b32", out, clk);
    run<23>("v_perm_b32", out, clk); run<24>("v_mov_b32", out, clk); run<25>("v_cmp_le_f32 vcc", out, clk); run<26>("v_cndmask_b32", out, clk);
    run<27>("v_cmp_lt_u32 sgpr", out, clk); run<28>("v_rcp_f32", out, clk); run<29>("v_sqrt_f32", out, clk); run<30>("v_cvt_f32_u32", out, clk);
    run<31>("v_div_fixup_f32", out, clk);
    printf("LDS (per 4 loads + wait, i.e. divide by 4 for one load; counted as 32 'instructions' per iteration):\n");
    run<32>("ds_read_b64 linear", out, clk); run<33>("node gather 3xb64+b32", out, clk); run<34>("node gather 4xb128", out, clk);
    return 0;
}
