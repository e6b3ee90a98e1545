// Microbenchmark: does a wave64 VALU instruction cost less when one 32-lane half is inactive?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, int iters, int mode) {
    const int lane = threadIdx.x & 63;
    bool act = mode == 0 ? true : mode == 1 ? lane < 32 : mode == 2 ? (lane & 1) == 0 : lane < 16;
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1, e = a + 2, f = a + 3;
    if (act) {
        for (int i = 0; i < iters; ++i) {
            a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c;
            a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f;
}
int main() {
    float *o; hipMalloc(&o, 256 * 8 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        hipEventRecord(s);
        hipLaunchKernelGGL(k, dim3(256 * 8), dim3(256), 0, 0, o, 200000, mode);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        printf("mode %d (%s): %.3f ms\n", mode, mode == 0 ? "all 64" : mode == 1 ? "lanes 0-31" : mode == 2 ? "even lanes" : "lanes 0-15", ms);
    }
    return 0;
}
