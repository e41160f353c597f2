// Randomised check on the GPU of a 3-instruction exact division by a divisor whose correctly rounded reciprocal is known:
//     y = RN(1 / a);  q0 = n * y;  r = fma(-a, q0, n);  q = fma(r, y, q0)          (Markstein's correction step)
// against the compiler's IEEE division n / a (-fhip-fp32-correctly-rounded-divide-sqrt), for |a| in [2^-30, 2^30] (either sign) and
// |n| in [2^-90, 2^90] — the ranges the render kernels guard — with uniformly random significands plus the patterns that
// stress rounding (all ones, all zeros, single bits, near-ties built from q * a).
//   hipcc -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 tools/micro/div_exact.hip -o tools/micro/div_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t rot(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
struct Rng { uint32_t s0, s1, s2, s3; };
__device__ __forceinline__ uint32_t next(Rng &g) {
    const uint32_t r = g.s0 + g.s3, t = g.s1 << 9;
    g.s2 ^= g.s0; g.s3 ^= g.s1; g.s1 ^= g.s2; g.s0 ^= g.s3; g.s2 ^= t; g.s3 = rot(g.s3, 11);
    return r;
}
__device__ __forceinline__ float recip_newton(float x) { const float r0 = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0); }
__device__ __forceinline__ float div_fast(float n, float a, float y) { const float q0 = n * y; return __builtin_fmaf(__builtin_fmaf(-a, q0, n), y, q0); }
__device__ __forceinline__ uint32_t mant(Rng &g, uint32_t kind) {
    const uint32_t r = next(g) & 0x7FFFFFu;
    switch (kind & 7u) {
    case 0: case 1: case 2: return r;                       // uniform
    case 3: return 0x7FFFFFu >> (r % 23u);                 // low bits all ones
    case 4: return 0x7FFFFFu & ~(0x7FFFFFu >> (r % 23u));  // high bits all ones
    case 5: return 1u << (r % 23u);                        // one bit
    case 6: return r & ~0xFFFu;                            // short significand
    default: return 0x7FFFFFu - (r & 0xFu);                // just below a power of two
    }
}

__global__ void k(unsigned long long *stats, uint32_t iters, uint32_t salt) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Rng g = {tid * 2654435761u + 1u, salt ^ 0x9E3779B9u, tid ^ 0x85EBCA6Bu, 0xC2B2AE35u + salt};
    for (int i = 0; i < 8; i++) next(g);
    unsigned long long bad = 0, done = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t k1 = next(g);
        const int ea = 127 - 30 + (int)(next(g) % 61u);          // a in [2^-30, 2^30]
        const float a = __uint_as_float(((uint32_t)ea << 23) | mant(g, k1));
        float n;
        if ((k1 >> 8) & 1u) {  // a numerator close to q * a for a random q: quotients near representable values and ties
            const float q = __uint_as_float(((uint32_t)(127 - 40 + (int)(next(g) % 81u)) << 23) | mant(g, k1 >> 3));
            const float p = q * a;
            n = __uint_as_float(__float_as_uint(p) + (next(g) % 5u) - 2u);
        } else {
            const int en = 127 - 90 + (int)(next(g) % 181u);     // |n| in [2^-90, 2^90]
            n = __uint_as_float(((uint32_t)en << 23) | mant(g, k1 >> 3));
        }
        if ((k1 >> 9) & 1u) n = -n;
        const float a_signed = ((k1 >> 10) & 1u) ? -a : a;  // the shade kernel divides by radii, which may be negative
        const float an = fabsf(n);
        if (!(an >= 0x1p-90f && an <= 0x1p90f)) continue;        // outside the guarded range
        const float y = recip_newton(a_signed);
        const float want = n / a_signed, got = div_fast(n, a_signed, y);
        done++;
        if (__float_as_uint(want) != __float_as_uint(got)) bad++;
    }
    atomicAdd(&stats[0], done);
    if (bad) atomicAdd(&stats[1], bad);
}

int main() {
    unsigned long long *d, h[2] = {0, 0};
    hipMalloc(&d, sizeof h);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (uint32_t pass = 0; pass < 8; pass++) {
        k<<<65536, 256>>>(d, 1024, pass * 7919u + 13u);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%llu divisions checked (|a| in [2^-30, 2^30], |n| in [2^-90, 2^90]): %llu differ from n / a\n", h[0], h[1]);
    return h[1] ? 2 : 0;
}
