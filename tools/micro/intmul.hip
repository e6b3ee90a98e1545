// Microbenchmark: issue cost of v_mul_lo_u32 against full-rate VALU, and of wang_hash (include/random_utils.h:7-14)
// with its 32-bit multiply by 0x27d4eb2d done by v_mul_lo_u32 or by three 16-bit multiply-adds (v_mad_u32_u16).
// 6 waves per SIMD, as the trace kernel runs.   hipcc --offload-arch=gfx950 -O3 -o /tmp/intmul tools/micro/intmul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mul_k16(uint32_t a) {
    // a * 0x27d4eb2d mod 2^32 = a.lo*K.lo + ((a.lo*K.hi + a.hi*K.lo) << 16)
    uint32_t t1, t2, t3;
    asm volatile("v_mad_u32_u16 %0, %1, %2, 0" : "=v"(t1) : "v"(a), "s"(0x27d4u));
    asm volatile("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(t2) : "v"(a), "s"(0xeb2du), "v"(t1));
    asm volatile("v_mad_u32_u16 %0, %1, %2, 0" : "=v"(t3) : "v"(a), "s"(0xeb2du));
    return (t2 << 16) + t3;
}
template <int kMode> __device__ __forceinline__ uint32_t hash(uint32_t s) {
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s ^= s >> 4;
    s = kMode == 1 ? mul_k16(s) : s * 0x27d4eb2du;
    s ^= s >> 15;
    return s;
}
template <int kMode> __global__ void __launch_bounds__(768, 2) k(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + blockIdx.x, b = a ^ 0x9e3779b9u, c = a + 77u, d = a * 3u;
    for (int i = 0; i < iters; ++i) {
        if (kMode == 0 || kMode == 1) { a = hash<kMode>(a); b = hash<kMode>(b); c = hash<kMode>(c); d = hash<kMode>(d); }
        else if (kMode == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { a *= 0x27d4eb2du; b *= 0x27d4eb2du; c *= 0x27d4eb2du; d *= 0x27d4eb2du; }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) { a = (a << 3) + b; b = (b << 3) + c; c = (c << 3) + d; d = (d << 3) + a; }      // v_lshl_add_u32
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int kMode> static void run(const char *what, uint32_t *o, int iters, int per_iter) {
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        hipEventRecord(s);
        hipLaunchKernelGGL(k<kMode>, dim3(512), dim3(768), 0, 0, o, iters);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        // 512 blocks x 12 waves on 1024 SIMDs = 6 waves per SIMD; cycles per wave-level unit at 2.4 GHz
        if (rep) printf("%-34s %.3f ms  = %.2f cycles per %s per SIMD\n", what, ms, ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * 6), kMode < 2 ? "hash" : "instruction");
    }
}
int main() {
    uint32_t *o; hipMalloc(&o, 512 * 768 * 4);
    uint32_t h0, h1;
    {   // correctness of the 16-bit form on the host side of things: compare device results
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(768), 0, 0, o, 5); hipMemcpy(&h0, o + 5, 4, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(768), 0, 0, o, 5); hipMemcpy(&h1, o + 5, 4, hipMemcpyDeviceToHost);
        printf("hash forms agree: %s (%08x %08x)\n", h0 == h1 ? "yes" : "NO", h0, h1);
    }
    run<3>("v_lshl_add_u32", o, 20000, 32);
    run<2>("v_mul_lo_u32", o, 20000, 32);
    run<0>("wang_hash, v_mul_lo_u32", o, 50000, 4);
    run<1>("wang_hash, 3 x v_mad_u32_u16", o, 50000, 4);
    return 0;
}
