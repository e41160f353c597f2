// Exhaustive check on the GPU: for which f32 inputs do short reciprocal sequences equal the correctly rounded 1.0f / x
// (the compiler's IEEE division: -fhip-fp32-correctly-rounded-divide-sqrt, no fast-math)?
//   hipcc -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 tools/micro/recip_exact.hip -o tools/micro/recip_exact
// Candidates: c1 = rcp + one fma-Newton step, c2 = two steps, c3 = three steps (the compiler's own chain without the scaling
// and fix-up instructions).  Reported per candidate: inputs where it differs from 1/x, split into "result is NaN" (a cheap
// guard catches those) and "result is a wrong number" (no guard possible), plus the exponent range of the wrong ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float nr(float x, float r) { const float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }

__global__ void k(unsigned long long *stats) {
    // stats per candidate c (0..2): [c*6+0] differing, [c*6+1] differing with NaN result, [c*6+2] wrong number,
    // [c*6+3] min biased exponent of wrong inputs, [c*6+4] max biased exponent of wrong inputs, [c*6+5] wrong among normal inputs in [2^-126, 2^126]
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long diff[3] = {0, 0, 0}, nan[3] = {0, 0, 0}, wrong[3] = {0, 0, 0}, wrong_mid[3] = {0, 0, 0};
    uint32_t emin[3] = {255, 255, 255}, emax[3] = {0, 0, 0};
    for (uint32_t j = 0; j < 256; j++) {
        const uint32_t bits = tid * 256u + j;
        const float x = __uint_as_float(bits);
        const float want = 1.0f / x;
        const float r0 = __builtin_amdgcn_rcpf(x);
        float c[3];
        c[0] = nr(x, r0);
        c[1] = nr(x, c[0]);
        // the compiler's chain: r1 = nr(r0); q0 = r1; q1 = q0 + (1 - x q0) r1; q2 = q1 + (1 - x q1) r1
        const float r1 = c[0];
        const float q1 = __builtin_fmaf(__builtin_fmaf(-x, r1, 1.0f), r1, r1);
        c[2] = __builtin_fmaf(__builtin_fmaf(-x, q1, 1.0f), r1, q1);
        const uint32_t e = (bits >> 23) & 255u;
        for (int i = 0; i < 3; i++) {
            const bool same = __float_as_uint(c[i]) == __float_as_uint(want) || (want != want && c[i] != c[i]);
            if (!same) {
                diff[i]++;
                if (c[i] != c[i]) nan[i]++;
                else { wrong[i]++; emin[i] = min(emin[i], e); emax[i] = max(emax[i], e); if (e >= 1 && e <= 253) wrong_mid[i]++; }
            }
        }
    }
    for (int i = 0; i < 3; i++) {
        if (diff[i]) atomicAdd(&stats[i * 6 + 0], diff[i]);
        if (nan[i]) atomicAdd(&stats[i * 6 + 1], nan[i]);
        if (wrong[i]) { atomicAdd(&stats[i * 6 + 2], wrong[i]); atomicMin(&stats[i * 6 + 3], (unsigned long long)emin[i]); atomicMax(&stats[i * 6 + 4], (unsigned long long)emax[i]); }
        if (wrong_mid[i]) atomicAdd(&stats[i * 6 + 5], wrong_mid[i]);
    }
}

int main() {
    unsigned long long *d, h[18];
    hipMalloc(&d, sizeof h);
    for (int i = 0; i < 18; i++) h[i] = (i % 6 == 3) ? 255 : 0;
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<65536, 256>>>(d);  // 2^24 threads x 256 inputs
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[3] = {"rcp + 1 Newton step", "rcp + 2 Newton steps", "compiler's chain without scaling / fix-up"};
    for (int i = 0; i < 3; i++)
        printf("%-45s differs on %llu inputs: %llu give NaN, %llu give a wrong number (biased exponent of x %llu..%llu; %llu of them with exponent 1..253)\n",
               names[i], h[i * 6], h[i * 6 + 1], h[i * 6 + 2], h[i * 6 + 3], h[i * 6 + 4], h[i * 6 + 5]);
    return 0;
}
