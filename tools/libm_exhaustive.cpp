// Dev tool (CPU): the exhaustive comparisons behind rt_device_math.h's restatements of the host libm's expf, powf(x, 5),
// acosf, atanf and atan2f.  Build and run (8 threads, about a minute):
//   g++ -O2 -ffp-contract=off -o /tmp/libm_exhaustive tools/libm_exhaustive.cpp -lpthread -lm && /tmp/libm_exhaustive
// (tests/test_device_math.py runs strided versions of the same sweeps on every CPU test run.)
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../ray-tracing-practice_amd/csrc/rt_device_math.h"

static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline bool same(float a, float b) { return fbits(a) == fbits(b) || (a != a && b != b); }
static float (*volatile libm_powf)(float, float) = powf;      // the library routine, not a compiler expansion

template <class F>
static void sweep(const char *what, uint64_t lo, uint64_t hi, F differs) {
    const int T = 8;
    std::atomic<long> bad{0}, n{0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            long b = 0, c = 0;
            for (uint64_t u = lo + t; u <= hi; u += T) { b += differs((uint32_t)u); ++c; }
            bad += b; n += c;
        });
    for (auto &t : th) t.join();
    std::printf("%-44s %12ld values, %ld differ\n", what, (long)n, (long)bad);
}

int main() {
    sweep("expf, all floats of [-128, -0]", 0x80000000u, 0xC3000000u, [](uint32_t u) { const float x = bitsf(u); return !same(rtd::exp_libm(x), expf(x)); });
    sweep("expf, all floats of [0, 88]", 0u, 0x42B00000u, [](uint32_t u) { const float x = bitsf(u); return !same(rtd::exp_libm(x), expf(x)); });
    sweep("powf(x, 5), all floats of [0, 2]", 0u, 0x40000000u, [](uint32_t u) { const float x = bitsf(u); return !same(rtd::pow5(x), libm_powf(x, 5.0f)); });
    for (int w = 2; w <= 6; w += 4) {
        char what[96];
        std::snprintf(what, sizeof what, "powf(x, 5) more than %d steps from pow5_float(x)", w);
        sweep(what, 0u, 0x40000000u, [w](uint32_t u) { const float x = bitsf(u);
            const int64_t d = (int64_t)fbits(rtd::pow5_float(x)) - (int64_t)fbits(libm_powf(x, 5.0f)); return d > w || d < -w; });
    }
    sweep("acosf, all floats of [-1, 1] (and NaN beyond)", 0u, 0xFFFFFFFFu, [](uint32_t u) { const float x = bitsf(u); return !same(rtd::acos_libm(x), acosf(x)); });
    sweep("atanf, all 2^32 floats", 0u, 0xFFFFFFFFu, [](uint32_t u) { const float x = bitsf(u); return !same(rtd::atan_libm(x), atanf(x)); });
    sweep("atan2f, 2^31 pairs (random bits / unit vectors)", 0u, 0x7FFFFFFFu, [](uint32_t u) {
        uint32_t h = rtd::wang_hash(u * 2654435761u + 12345u);
        const uint32_t a = h;
        h = rtd::wang_hash(h ^ 0x9e3779b9u);
        const uint32_t b = h;
        float y = bitsf(a), x = bitsf(b);
        if (u & 1) { y = (float)((int32_t)a) * 4.6566e-10f; x = (float)((int32_t)b) * 4.6566e-10f; if (u & 2) y *= 1e-3f; if (u & 4) x *= 1e-4f; }
        return !same(rtd::atan2_libm(y, x), atan2f(y, x)); });
    return 0;
}
