// Exhaustive check on the device: for EVERY binary32 input, is a short v_rcp_f32 / v_rsq_f32 + fused-multiply-add sequence
// bit-identical to the correctly rounded 1.0f / x and sqrtf(x) the compiler emits under -fhip-fp32-correctly-rounded-divide-sqrt
// (58 and 62 cycles of SIMD time each, tools/micro/valu_rates.hip)?  Prints the mismatch count per candidate and per
// exponent, so that a cheap range test can fence off whatever does not match.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -o tools/micro/exact_rcp_sqrt.bin tools/micro/exact_rcp_sqrt.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float rcp_a(float x) { const float r = __builtin_amdgcn_rcpf(x); const float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float rcp_b(float x) {
    float r = __builtin_amdgcn_rcpf(x); float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float sqrt_a(float x) {
    const float r = __builtin_amdgcn_rsqf(x); const float s = x * r, h = 0.5f * r; const float e = __builtin_fmaf(-s, s, x); return __builtin_fmaf(e, h, s);
}
__device__ __forceinline__ float sqrt_b(float x) {
    const float s = __builtin_amdgcn_sqrtf(x); const float h = 0.5f * __builtin_amdgcn_rsqf(x); const float e = __builtin_fmaf(-s, s, x); return __builtin_fmaf(e, h, s);
}
__device__ __forceinline__ float sqrt_c(float x) {     // two corrections
    const float r = __builtin_amdgcn_rsqf(x); float s = x * r; const float h = 0.5f * r; float e = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(e, h, s);
    e = __builtin_fmaf(-s, s, x); return __builtin_fmaf(e, h, s);
}
// counts[cand][exponent]: mismatches;  cand 0,1: rcp_a, rcp_b;  2,3,4: sqrt_a, sqrt_b, sqrt_c
__global__ void check(unsigned long long *counts, uint32_t *example) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const uint32_t bits = (uint32_t)b;
        const float x = __uint_as_float(bits);
        if (x != x) continue;                     // NaN inputs are not compared
        const uint32_t ex = (bits >> 23) & 255u;
        const float want = 1.0f / x;
        const float ga = rcp_a(x), gb = rcp_b(x);
        if (__float_as_uint(ga) != __float_as_uint(want) && !(ga != ga && want != want)) { atomicAdd(&counts[0 * 256 + ex], 1ull); example[0] = bits; }
        if (__float_as_uint(gb) != __float_as_uint(want) && !(gb != gb && want != want)) { atomicAdd(&counts[1 * 256 + ex], 1ull); example[1] = bits; }
        if (!(bits >> 31)) {
            const float ws = sqrtf(x);
            const float sa = sqrt_a(x), sb = sqrt_b(x), sc = sqrt_c(x);
            if (__float_as_uint(sa) != __float_as_uint(ws) && !(sa != sa && ws != ws)) { atomicAdd(&counts[2 * 256 + ex], 1ull); example[2] = bits; }
            if (__float_as_uint(sb) != __float_as_uint(ws) && !(sb != sb && ws != ws)) { atomicAdd(&counts[3 * 256 + ex], 1ull); example[3] = bits; }
            if (__float_as_uint(sc) != __float_as_uint(ws) && !(sc != sc && ws != ws)) { atomicAdd(&counts[4 * 256 + ex], 1ull); example[4] = bits; }
        }
    }
}
int main() {
    unsigned long long *c; uint32_t *ex;
    (void)hipMalloc(&c, 5 * 256 * 8); (void)hipMemset(c, 0, 5 * 256 * 8); (void)hipMalloc(&ex, 5 * 4); (void)hipMemset(ex, 0, 20);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, c, ex);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    static unsigned long long h[5 * 256]; uint32_t he[5];
    (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(he, ex, sizeof he, hipMemcpyDeviceToHost);
    const char *names[5] = {"rcp + 1 correction", "rcp + 2 corrections", "rsq, x*r + 1 correction", "sqrt + 1 correction (rsq/2)", "rsq, x*r + 2 corrections"};
    for (int k = 0; k < 5; ++k) {
        unsigned long long tot = 0; int lo = 256, hi = -1;
        for (int e = 0; e < 256; ++e) if (h[k * 256 + e]) { tot += h[k * 256 + e]; if (e < lo) lo = e; if (e > hi) hi = e; }
        float exf; memcpy(&exf, &he[k], 4);
        printf("%-30s mismatches %llu", names[k], tot);
        if (tot) {
            printf("  biased exponents with mismatches: %d..%d, example input %08x (%g);", lo, hi, he[k], exf);
            unsigned long long mid = 0; for (int e = 32; e <= 222; ++e) mid += h[k * 256 + e];
            printf(" of them with exponent in [32, 222]: %llu;", mid);
            printf(" exponent ranges with mismatches:");
            for (int e = 0; e < 256;) { if (!h[k * 256 + e]) { ++e; continue; } int f = e; while (f + 1 < 256 && h[k * 256 + f + 1]) ++f; printf(" %d-%d", e, f); e = f + 1; }
        }
        printf("\n");
    }
    return 0;
}
