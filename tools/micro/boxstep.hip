// Microbenchmark: SIMD issue cost of the box-step arithmetic with the node record in registers,
// measured in shader cycles (s_memtime) so the clock does not matter.  blocks*4 waves spread over
// 1024 SIMDs; cycles are per loop iteration per resident wave (divide by waves/SIMD for SIMD time).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float pk2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters) {
    const int lane = threadIdx.x;
    float ox = lane * 0.01f, oy = 0.3f, oz = -0.2f, ix = 1.1f, iy = -0.7f, iz = 0.9f, closest = 1e30f;
    float4 A = make_float4(-1.f, 1.f, -1.5f, 1.2f), B = make_float4(-0.5f, 0.8f, 0.f, 0.f);
    int node = lane, sp = 0;
    const bool sx = out[0] < 0, sy = out[1] < 0, sz = out[2] < 0;     // unknown at compile time
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        float tmin, tmax;
        if (MODE == 0) {
            const pk2 tx = (pk2{A.x, A.y} - pk2{ox, ox}) * pk2{ix, ix};
            const pk2 ty = (pk2{A.z, A.w} - pk2{oy, oy}) * pk2{iy, iy};
            const pk2 tz = (pk2{B.x, B.y} - pk2{oz, oz}) * pk2{iz, iz};
            const float nx = sx ? tx.y : tx.x, fx = sx ? tx.x : tx.y;
            const float ny = sy ? ty.y : ty.x, fy = sy ? ty.x : ty.y;
            const float nz = sz ? tz.y : tz.x, fz = sz ? tz.x : tz.y;
            tmin = fmaxf(fmaxf(fmaxf(nx, ny), nz), 0.001f);
            tmax = fminf(fminf(fminf(fx, fy), fz), closest);
        } else {
            const float a = (A.x - ox) * ix, b = (A.y - ox) * ix, c = (A.z - oy) * iy, d = (A.w - oy) * iy, e = (B.x - oz) * iz, f = (B.y - oz) * iz;
            const float nx = sx ? b : a, fx = sx ? a : b, ny = sy ? d : c, fy = sy ? c : d, nz = sz ? f : e, fz = sz ? e : f;
            tmin = fmaxf(fmaxf(fmaxf(nx, ny), nz), 0.001f);
            tmax = fminf(fminf(fminf(fx, fy), fz), closest);
        }
        const bool h = tmax > tmin;
        sp = h ? node : 0;
        node = h ? node + 1 : node + 3;
        // feed results back so nothing is loop invariant (cheap: 1 op each)
        A.x = __int_as_float(__float_as_int(A.x) ^ (node & 1));
        A.z = __int_as_float(__float_as_int(A.z) ^ (sp & 1));
        B.x = __int_as_float(__float_as_int(B.x) ^ (node & 2));
        ox = __int_as_float(__float_as_int(ox) ^ (sp & 2));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[4 + blockIdx.x * blockDim.x + threadIdx.x] = A.x + A.z + B.x + ox + node + sp;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name, float *o, unsigned long long *c, int blocks) {
    const int iters = 100000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, c, 1000);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipEventRecord(s);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, c, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double wps = blocks * 4.0 / 1024;
    printf("%-18s %2.0f waves/SIMD: wall %.3f ms = %.1f ns per wave-iteration -> %.2f ns of SIMD time; s_memtime ticks/iter %.1f (tick = %.2f ns)\n", name, wps, ms,
           ms * 1e6 / iters, ms * 1e6 / iters / wps, (double)h / iters, ms * 1e6 / iters / ((double)h / iters));
}
int main() {
    float *o; unsigned long long *c; hipMalloc(&o, 256 * 64 * 256 * 4 + 64); hipMalloc(&c, 8); hipMemset(o, 0, 64);
    for (int blocks : {256, 1024, 1536, 2048}) { run<0>("packed box step", o, c, blocks); run<1>("scalar box step", o, c, blocks); }
    return 0;
}
