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
//   * expf, powf(x, 5) (Schlick), acosf and atan2f (a textured sphere's uv) follow the algorithms of the
//     host libm the reference calls (glibc 2.35: e_expf.c and e_powf.c — table + polynomial in double, in the
//     FMA build x86-64 hosts run — and the fdlibm-derived float routines e_acosf.c, s_atanf.c, e_atan2f.c),
//     restated operation for operation and compared with that libm for every float of their domains
//     (tools/libm_exhaustive.cpp: 0 differences), so these match the CPU to the last bit as well.
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
    // / 4294967296.0f: an exact scaling, so ldexp gives the bits of the multiplication (include/random_utils.h:18).  On the
    // device it also keeps the constant 2^-32 out of the vector registers: as a multiplier the compiler paired it with another
    // product into one v_pk_mul_f32, and the register pair it pinned for that across the render kernel's main loop was the one
    // thing the 64-register build had to spill (reloaded in every glass shade step); v_ldexp_f32 takes its exponent from a
    // scalar register.
    return __builtin_ldexpf((float)seed, -32);
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
    // The build of this routine that x86-64 hosts with FMA run (glibc's ifunc picks __expf_fma) has `z - kd` and the
    // polynomial contracted to fused multiply-adds; with exactly these four fusions the values below are that libm's
    // for EVERY float in [-128, 0] and [0, 88] (2.24e9 values, tests/test_device_math.py; the unfused form differs on 2).
    const double r = __builtin_fma(inv_ln2_n, xd, -kd);
    uint64_t t = exp2_tab_entry((uint32_t)(ki % 32u));
    t += ki << (52 - 5);
    double s;
    memcpy(&s, &t, 8);
    z = __builtin_fma(c0, r, c1);
    const double r2 = r * r;
    double y = __builtin_fma(c2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

// ---- powf(x, 5): the host libm's algorithm ----------------------------------------------------------
// (1 - cos)^5 of the Schlick term (include/materials.h:67).  glibc >= 2.28 sysdeps/ieee754/flt-32/e_powf.c: log2(x) from a
// 16-entry table (1/c, log2 c) and a degree-5 polynomial, times y, then 2^(.) with the table and cubic exp_libm uses — all in
// double.  As with expf, the build x86-64 hosts with FMA run has every a*b+c fused; with those fusions the value below is
// that libm's powf(x, 5.0f) for EVERY float in [2^-126, 2] (1.07e9 values, tests/test_device_math.py; unfused: 3 differ).
// A correctly rounded x^5 — round 2's stand-in — differs from it by one ulp on 0.0136 % of [0, 1].
#if defined(__HIPCC__)
__device__ __constant__
#endif
static const double kPowLog2Tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
RT_HD float pow5(float x) {
    uint32_t ix;
    memcpy(&ix, &x, 4);
    bool negative = false;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {            // x < 2^-126, negative, inf or NaN
        if ((ix << 1) == 0u || (ix << 1) >= 0xff000000u) {           // zero, inf, NaN: x * x with the sign of x (5 is odd)
            const float x2 = x * x;
            return (ix & 0x80000000u) ? -x2 : x2;
        }
        if (ix & 0x80000000u) { negative = true; ix &= 0x7fffffffu; }
        if (ix < 0x00800000u) {                                       // subnormal: normalise, the exponent goes negative
            const float ax = fabsf(x) * 0x1p23f;
            memcpy(&ix, &ax, 4);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline: x = 2^k z, z in [0x1.66p-1, 0x1.66p0)
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) % 16u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int32_t k = (int32_t)top >> 23;
    float zf;
    memcpy(&zf, &iz, 4);
    const double invc = kPowLog2Tab[i][0], logc = kPowLog2Tab[i][1], z = (double)zf;
    const double a0 = 0x1.27616c9496e0bp-2, a1 = -0x1.71969a075c67ap-2, a2 = 0x1.ec70a6ca7baddp-2, a3 = -0x1.7154748bef6c8p-1,
                 a4 = 0x1.71547652ab82bp0;
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = __builtin_fma(a0, r, a1);
    const double p = __builtin_fma(a2, r, a3);
    const double r4 = r2 * r2;
    double q = __builtin_fma(a4, r, y0);
    q = __builtin_fma(p, r2, q);
    y = __builtin_fma(y, r4, q);
    const double ylogx = 5.0 * y;
    uint64_t yb;
    memcpy(&yb, &ylogx, 8);
    if (((yb >> 47) & 0xffffu) >= (0x405f800000000000ull >> 47)) {       // |5 log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return negative ? -INFINITY : INFINITY;
        if (ylogx <= -150.0) return negative ? -0.0f : 0.0f;
    }
    // exp2_inline: 2^(k/32) from the table, 2^r by the cubic
    const double shift = 0x1.8p+52 / 32.0;
    double kd = ylogx + shift;
    uint64_t ki;
    memcpy(&ki, &kd, 8);
    kd -= shift;
    const double rr = ylogx - kd;
    uint64_t t = exp2_tab_entry((uint32_t)(ki % 32u));
    t += ki << (52 - 5);
    double s;
    memcpy(&s, &t, 8);
    const double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double yy = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    yy = __builtin_fma(zz, rr2, yy);
    yy = yy * s;
    const float res = (float)yy;
    return negative ? -res : res;
}

// ---- acosf, atanf, atan2f: the host libm's algorithms ---------------------------------------------------
// get_sphere_uv (include/sphere.h:16-22) calls acosf and atan2f of the host libm on the CPU path that is the oracle.
// glibc 2.35 still carries the float conversions of Sun's fdlibm for these (sysdeps/ieee754/flt-32/e_acosf.c,
// s_atanf.c, e_atan2f.c: rational / polynomial approximations in plain float arithmetic), restated here operation for
// operation.  Compiled without contraction (the kernel's and the tests' flags) they return that libm's bits for ALL floats
// in [-1, 1] (acosf), for all 2^32 floats (atanf), and on 1.2e9 random and unit-vector-like pairs (atan2f):
// tests/test_device_math.py.  sqrt and the divisions are single correctly rounded operations on both sides.
RT_HD float acos_libm(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    uint32_t ux;
    memcpy(&ux, &x, 4);
    const int32_t hx = (int32_t)ux, ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;        // |x| == 1
    if (ix > 0x3f800000) return (x - x) / (x - x);                               // |x| > 1: NaN
    if (ix < 0x3f000000) {                                                       // |x| < 0.5
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {                                                                // x < -0.5
        const float z = (one + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrt_cr(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (one - x) * 0.5f;                                            // x > 0.5
    const float s = sqrt_cr(z);
    uint32_t us;
    memcpy(&us, &s, 4);
    us &= 0xfffff000u;
    float df;
    memcpy(&df, &us, 4);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}
RT_HD float atan_libm(float x) {
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const float one = 1.0f;
    uint32_t ux;
    memcpy(&ux, &x, 4);
    const int32_t hx = (int32_t)ux, ix = hx & 0x7fffffff;
    int id;
    float hi = 0.0f, lo = 0.0f;
    if (ix >= 0x4c000000) {                                                      // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;                                       // NaN
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x3ee00000) {                                                       // |x| < 0.4375
        if (ix < 0x31000000) return x;                                           // |x| < 2^-29
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {                                                   // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; hi = hi0; lo = lo0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; hi = hi1; lo = lo1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; hi = hi2; lo = lo2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; hi = hi3; lo = lo3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}
RT_HD float atan2_libm(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    uint32_t ux, uy;
    memcpy(&ux, &x, 4);
    memcpy(&uy, &y, 4);
    const int32_t hx = (int32_t)ux, hy = (int32_t)uy, ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;                        // NaN
    if (hx == 0x3f800000) return atan_libm(y);                                   // x == 1
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                       // 2 sign(x) + sign(y)
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);           // y == +-0
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;                 // x == +-0
    if (ix == 0x7f800000) {                                                      // x infinite
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;        // y infinite
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                                       // |y / x| > 2^60
    else if (hx < 0 && k < -60) z = 0.0f;                                        // |y| / x < -2^60
    else z = atan_libm(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// Schlick's r0^2 for a refraction ratio (the first two lines of reflectance()): made once per material and side by the
// packer (rt_accel.cpp keeps both in the albedo slots a DIELECTRIC never reads), so the kernel's glass branch has no division.
RT_HD float schlick_r0sq(float ref_idx) {
    const float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    return r0 * r0;
}
RT_HD float reflectance_from_r0sq(float cosine, float r0) { return r0 + (1.0f - r0) * pow5(1.0f - cosine); }
// x^5 by three float multiplications: at most 4 x 2^-24 away from x^5 in relative terms, i.e. within 4 ulps of it (and within
// one subnormal step where the result is subnormal); the host libm's powf(x, 5) is within 1.5 ulps of x^5.  So the libm's
// value is never more than kPow5Window = 6 float steps away from this one — checked for every float of [0, 2]
// (tools/libm_exhaustive.cpp: 0 of 1.07e9 beyond 6 steps, 65 k beyond 2), which is all the next function needs.
RT_HD float pow5_float(float x) {
    const float x2 = x * x;
    return (x2 * x2) * x;
}
constexpr uint32_t kPow5Window = 6;
// reflectance(cosine, ·) > rnd (include/materials.h:108), decided exactly as with the host libm's powf but without
// evaluating it: r0 + (1 - r0) * p is a non-decreasing function of p in float arithmetic (1 - r0 > 0, rounding is monotonic),
// and the libm's p lies within kPow5Window steps of pow5_float(x).  If even the lower end of that window gives a value above
// rnd the answer is yes, if not even the upper end does it is no; only when rnd falls inside the window (~1e-6 of the
// draws) is powf itself restated (pow5) — by the exact walk: the guarded trace kernel hands such a sample to the re-walk launch
// like any other it cannot vouch for, so its hot shade step carries neither the table-driven code nor its registers, nor any
// double-precision temporaries (the glass branch runs in nine shade steps of ten; evaluating pow5 there cost 8 % of the
// headline frame).
// schlick_bracket: 1 = yes, 0 = no, -1 = rnd is inside the window (or the argument outside [0, 2]).
RT_HD int schlick_bracket(float cosine, float r0, float rnd) {
    const float x = 1.0f - cosine;
    const float p = pow5_float(x);
    uint32_t pb;
    memcpy(&pb, &p, 4);
    const uint32_t lb = pb > kPow5Window ? pb - kPow5Window : 0u, hb = pb + kPow5Window;       // (x in [0, 2]: p >= 0, its bits are its rank)
    float lo, hi;
    memcpy(&lo, &lb, 4);
    memcpy(&hi, &hb, 4);
    const float k = 1.0f - r0;
    const bool surely = (r0 + k * lo) > rnd;
    const bool maybe = (r0 + k * hi) > rnd;
    return (surely == maybe && x >= 0.0f && x <= 2.0f) ? (surely ? 1 : 0) : -1;
}
RT_HD bool schlick_exceeds(float cosine, float r0, float rnd) {
    const int b = schlick_bracket(cosine, r0, rnd);
    if (b >= 0) return b != 0;
    return (r0 + (1.0f - r0) * pow5(1.0f - cosine)) > rnd;
}
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
