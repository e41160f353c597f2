// Exhaustive check on the GPU: for which positive f32 inputs do short square-root sequences equal the correctly rounded
// sqrtf(x) (the compiler's IEEE square root: -fhip-fp32-correctly-rounded-divide-sqrt, no fast-math)?
//   hipcc -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 tools/micro/sqrt_exact.hip -o tools/micro/sqrt_exact
// Candidates: c0 = reciprocal-root estimate and one coupled Newton step (y = rsq x; s = x y; h = y / 2; s + (x - s s) h),
// c1 = the same with a second step, c2 = root estimate corrected with the reciprocal-root estimate
// (s = sqrt x; s + (x - s s) (rsq x / 2)).  Reported per candidate: inputs where it differs from sqrtf, split into
// "result is NaN" and "result is a wrong number", plus the exponent range of the wrong ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k(unsigned long long *stats) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long diff[3] = {0, 0, 0}, nan[3] = {0, 0, 0}, wrong[3] = {0, 0, 0}, wrong_mid[3] = {0, 0, 0};
    uint32_t emin[3] = {255, 255, 255}, emax[3] = {0, 0, 0};
    for (uint32_t j = 0; j < 128; j++) {
        const uint32_t bits = tid * 128u + j;  // 2^24 threads x 128 = every non-negative pattern
        const float x = __uint_as_float(bits);
        const float want = sqrtf(x);
        const float y = __builtin_amdgcn_rsqf(x);
        const float h = 0.5f * y;
        float c[3];
        const float s0 = x * y;
        c[0] = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), h, s0);
        c[1] = __builtin_fmaf(__builtin_fmaf(-c[0], c[0], x), h, c[0]);
        const float s = __builtin_amdgcn_sqrtf(x);
        c[2] = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
        const uint32_t e = (bits >> 23) & 255u;
        for (int i = 0; i < 3; i++) {
            const bool same = __float_as_uint(c[i]) == __float_as_uint(want) || (want != want && c[i] != c[i]);
            if (!same) {
                diff[i]++;
                if (c[i] != c[i]) nan[i]++;
                else { wrong[i]++; emin[i] = min(emin[i], e); emax[i] = max(emax[i], e); if (e >= 31 && e <= 223) wrong_mid[i]++; }
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
    k<<<65536, 256>>>(d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[3] = {"rsq, one coupled Newton step", "rsq, two coupled Newton steps", "sqrt corrected with rsq / 2"};
    for (int i = 0; i < 3; i++)
        printf("%-32s differs on %llu inputs: %llu give NaN, %llu give a wrong number (biased exponent of x %llu..%llu; %llu of them with exponent 31..223)\n",
               names[i], h[i * 6], h[i * 6 + 1], h[i * 6 + 2], h[i * 6 + 3], h[i * 6 + 4], h[i * 6 + 5]);
    return 0;
}
