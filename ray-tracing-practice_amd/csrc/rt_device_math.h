// rt_device_math.h — scalar/vector arithmetic of the render kernel.
//
// Everything here is written so that, compiled with -ffp-contract=off and IEEE divide/sqrt, a
// gfx950 lane computes bit for bit what the reference's CPU path computes on x86-64:
//   * float sums/products are evaluated in the reference's order, never fused;
//   * where the reference goes through double for ONE operation on float operands and narrows
//     the result (1.0/t in vec3 operator/, 1.0 - x before sqrtf/fabsf, the plane root), the plain
//     float operation is used instead: with 53 >= 2*24+2 significand bits the double rounding is
//     innocuous for + - * / sqrt, so both give the same float (tests/test_device_math.py checks
//     this on the CPU);
//   * where two double operations are chained before narrowing (sphere roots,
//     include/sphere.h:35-41; triangle a+b>1, include/plane.h:49) real fp64 is used;
//   * expf follows the algorithm of the host libm the reference calls (glibc >= 2.27
//     sysdeps/ieee754/flt-32/e_expf.c: 2^(k/32) table + cubic in double), so Beer-Lambert
//     transmission matches the CPU to the last bit; powf(x, 5) of the Schlick term is evaluated as
//     a correctly rounded x^5 (glibc's powf is within 1 ulp of that, see DESIGN.md "Numerics").
//
// The functions are plain inline C++ marked RT_HD so the same source is compiled for the device
// and — by tests only — for the host, where it is compared against libm.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD static inline
#endif

namespace rtd {

struct f3 { float x, y, z; };

RT_HD f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 mul(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 scale(float t, f3 v) { return mk(t * v.x, t * v.y, t * v.z); }
RT_HD f3 neg(f3 a) { return mk(-a.x, -a.y, -a.z); }
RT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // include/vec3.h:99
RT_HD float lensq(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }          // include/vec3.h:55
RT_HD f3 cross(f3 a, f3 b) {                                                   // include/vec3.h:101-103
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// ---- correctly rounded 1/t and sqrt(t) ---------------------------------------------------------
// On the host: the C operators.  On gfx950 the compiler's correctly rounded forms (-fhip-fp32-correctly-rounded-divide-sqrt:
// v_div_scale / v_rcp / four fma / v_div_fmas / v_div_fixup; v_sqrt plus a two-sided residual test) cost 58 and 62 cycles of
// SIMD time per wave, a v_fma_f32 2.8 (tools/micro/valu_rates.hip) — and a shade step holds a dozen of them.  One hardware
// estimate + one fused correction gives THE SAME BITS for every input whose exponent is not extreme:
//     1/t:      r = v_rcp_f32(t);  r += r * fma(-t, r, 1)                  for |t| in [2^-126, 2^126)
//     sqrt(t):  r = v_rsq_f32(t);  s = t*r;  s += (r/2) * fma(-s, s, t)    for  t  in [2^-102, 2^128)
// checked on the device for ALL 2^32 inputs (rt_debug_check_fast_math, tests/test_gpu_parity.py: zero mismatches with the
// range fence below; tools/micro/exact_rcp_sqrt.hip prints where the unfenced sequences differ: only biased exponents 0
// and 253-255 for 1/t, 0-24 and 255 for sqrt).  Inputs outside the fence — zero, denormals, infinities, NaN, huge — take
// the compiler's sequence in a branch the wave skips when no lane needs it.
#ifndef RTP_FAST_RCP_SQRT
#define RTP_FAST_RCP_SQRT 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && RTP_FAST_RCP_SQRT
__device__ __forceinline__ float recip_estimate(float t) {          // exact where recip_in_fence(t)
    const float r = __builtin_amdgcn_rcpf(t);
    return __builtin_fmaf(__builtin_fmaf(-t, r, 1.0f), r, r);
}
__device__ __forceinline__ bool recip_in_fence(float t) { return (__float_as_uint(t) & 0x7fffffffu) - 0x00800000u < 0x7e000000u; }
__device__ __forceinline__ float recip(float t) {
    float r = recip_estimate(t);
    if (__builtin_expect(!recip_in_fence(t), 0)) r = 1.0f / t;
    return r;
}
__device__ __forceinline__ float sqrt_cr(float t) {
    const float r = __builtin_amdgcn_rsqf(t);
    const float s0 = t * r, h = 0.5f * r;
    float s = __builtin_fmaf(__builtin_fmaf(-s0, s0, t), h, s0);
    if (__builtin_expect(!(__float_as_uint(t) - 0x0c800000u < 0x73000000u), 0)) s = sqrtf(t);
    return s;
}
#else
// (1.0 / t) narrowed to float == 1.0f / t (single correctly rounded operation, see header).
RT_HD float recip(float t) { return 1.0f / t; }
RT_HD float sqrt_cr(float t) { return sqrtf(t); }
#endif
RT_HD f3 divs(f3 v, float t) { return scale(recip(t), v); }                    // include/vec3.h:97
RT_HD f3 unit(f3 v) { return divs(v, sqrt_cr(lensq(v))); }                     // include/vec3.h:105
RT_HD bool near_zero(f3 a) {                                                   // include/vec3.h:58-61
    const float s = 1e-8f;
    return (fabsf(a.x) < s) && (fabsf(a.y) < s) && (fabsf(a.z) < s);
}
RT_HD f3 reflect(f3 v, f3 n) { return sub(v, scale(2.0f * dot(v, n), n)); }    // include/vec3.h:63
RT_HD f3 refract(f3 v, f3 n, float eta) {                                      // include/vec3.h:65-70
    const float cos_theta = fminf(dot(neg(v), n), 1.0f);
    const f3 perp = scale(eta, add(v, scale(cos_theta, n)));
    const f3 par = scale(-sqrt_cr(fabsf(1.0f - lensq(perp))), n);
    return add(perp, par);
}

// ---- RNG: include/random_utils.h:7-42 --------------------------------------------------------
RT_HD uint32_t wang_hash(uint32_t s) {
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s ^= s >> 4;
    s *= 0x27d4eb2du;
    s ^= s >> 15;
    return s;
}
RT_HD float random_float(uint32_t &seed) {      // may return exactly 1.0f
    seed = wang_hash(seed);
    return (float)seed * 2.3283064365386962890625e-10f;   // / 4294967296.0f, exact scaling
}
// min + (max - min) * r = -1 + 2 * r.  r = (float)seed * 2^-32 and 2 * r are exact (powers of two), so the one rounding of
// the sum is the one rounding of fma((float)seed, 2^-31, -1): two instructions per coordinate instead of four in the
// rejection-sampling loop, the longest-running loop of a shade step.
RT_HD float random_pm1(uint32_t &seed) {
    seed = wang_hash(seed);
    return __builtin_fmaf((float)seed, 4.656612873077392578125e-10f, -1.0f);
}
RT_HD f3 random_in_unit_sphere(uint32_t &seed) {
    for (;;) {
        const float x = random_pm1(seed);
        const float y = random_pm1(seed);
        const float z = random_pm1(seed);
        const f3 c = mk(x, y, z);
        if (lensq(c) < 1.0f) return c;
    }
}
RT_HD f3 random_in_hemisphere(f3 normal, uint32_t &seed) {
    const f3 s = unit(random_in_unit_sphere(seed));
    return dot(s, normal) > 0.0f ? s : neg(s);
}

// ---- expf: the host libm's algorithm -----------------------------------------------------------
// 2^(i/32) as IEEE doubles with i<<47 subtracted from the bit pattern (the exponent is added back
// from k): the layout of glibc's __exp2f_data.tab, regenerated from correctly rounded 2^(i/32).
#if defined(__HIPCC__)
__device__ __constant__
#endif
static const uint64_t kExp2Tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

RT_HD uint64_t exp2_tab_entry(uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return kExp2Tab[i];
#else
    return kExp2Tab[i];
#endif
}

// exp(x) for x <= 0 (the only range Beer-Lambert produces) and moderate positive x.
RT_HD float exp_libm(float x) {
    if (!(x >= -104.0f)) return x != x ? x : 0.0f;      // underflow to 0 (and NaN passthrough)
    if (x > 88.0f) return x * 3.0e38f;                  // overflow → +inf (unused by the renderer)
    const double N = 32.0;
    const double inv_ln2_n = 0x1.71547652b82fep+0 * N;
    const double shift = 0x1.8p+52;
    const double c0 = 0x1.c6af84b912394p-5 / N / N / N;
    const double c1 = 0x1.ebfce50fac4f3p-3 / N / N;
    const double c2 = 0x1.62e42ff0c52d6p-1 / N;
    const double xd = (double)x;
    double z = inv_ln2_n * xd;
    double kd = z + shift;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= shift;
    const double r = z - kd;
    uint64_t t = exp2_tab_entry((uint32_t)(ki % 32u));
    t += ki << (52 - 5);
    double s;
    memcpy(&s, &t, 8);
    z = c0 * r + c1;
    const double r2 = r * r;
    double y = c2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

// (1-cos)^5 of the Schlick term (include/materials.h:67): correctly rounded via double
// (x^2 exact, two more roundings at 2^-53 — invisible after narrowing except on a float tie).
RT_HD float pow5(float x) {
    const double d = (double)x;
    const double d2 = d * d;
    return (float)(d2 * d2 * d);
}

// Schlick's r0^2 for a refraction ratio (the first two lines of reflectance()): made once per material and side by the
// packer (rt_accel.cpp keeps both in the albedo slots a DIELECTRIC never reads), so the kernel's glass branch has no division.
RT_HD float schlick_r0sq(float ref_idx) {
    const float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    return r0 * r0;
}
RT_HD float reflectance_from_r0sq(float cosine, float r0) { return r0 + (1.0f - r0) * pow5(1.0f - cosine); }
RT_HD float reflectance(float cosine, float ref_idx) {     // include/materials.h:64-68
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * pow5(1.0f - cosine);
}

// Saver arithmetic: ISaver::writeColor (src/camera.cu:138-147) for one channel.
RT_HD uint8_t tonemap_u8(float sum, float inv_divisor) {
    const float g = sqrtf(inv_divisor * sum);
    float c = g;
    if (g < 0.0f) c = 0.0f;
    if (g > 0.999f) c = 0.999f;
    return (uint8_t)(int)(256.0f * c);   // NaN → 0, as the x86 conversion of the reference yields
}

}  // namespace rtd
