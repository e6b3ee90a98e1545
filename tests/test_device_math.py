"""CPU checks of ray-tracing-practice_amd/csrc/rt_device_math.h (the kernel's arithmetic),
compiled for the host: the float shortcuts it takes are exact, and its expf/pow5 agree with the
host libm the reference calls."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "%s/ray-tracing-practice_amd/csrc/rt_device_math.h"
#include <cmath>
extern "C" {
// returns number of mismatches over a strided sweep of float bit patterns
long sweep_exp(unsigned lo, unsigned hi, unsigned stride) {
    long bad = 0;
    for (unsigned long u = lo; u <= hi; u += stride) { unsigned b = (unsigned)u; float x; memcpy(&x, &b, 4);
        float a = rtd::exp_libm(x), g = expf(x); if (memcmp(&a, &g, 4)) bad++; }
    return bad;
}
long sweep_pow5(unsigned stride, long *total) {
    long bad = 0, n = 0;
    for (unsigned long u = 0; u <= 0x3F800000ul; u += stride) { unsigned b = (unsigned)u; float x; memcpy(&x, &b, 4);
        float a = rtd::pow5(x), g = powf(x, 5); if (memcmp(&a, &g, 4)) bad++; n++; }
    *total = n; return bad;
}
// single-operation-through-double identities the kernel relies on
long sweep_identities(unsigned stride) {
    long bad = 0;
    for (unsigned long u = 1; u < 0x7F800000ul; u += stride) { unsigned b = (unsigned)u; float x; memcpy(&x, &b, 4);
        float r1 = (float)(1.0 / (double)x), r2 = rtd::recip(x); if (memcmp(&r1, &r2, 4)) bad++;
        float y = x * 0.37f;
        float s1 = (float)(1.0 - (double)y), s2 = 1.0f - y; if (memcmp(&s1, &s2, 4)) bad++;
        float q1 = (float)((double)y / (double)x), q2 = y / x; if (memcmp(&q1, &q2, 4)) bad++;
        float h1 = (float)((double)y - 0.5), h2 = y - 0.5f; if (memcmp(&h1, &h2, 4)) bad++;
        float p1 = powf(y, 2), p2 = y * y; if (p1 == p1 && memcmp(&p1, &p2, 4)) bad++;
    }
    return bad;
}
// ELLIPSE interior test: the reference writes powf(x, 2) with a literal exponent (include/plane.h:41); the kernel
// evaluates x * x.  `literal`: the call as the reference writes it, which gcc/clang/nvcc at -O1 and above expand to
// x * x (both of the reference's build files use -O3; the oracle is built with -O2); `libm`: the library routine
// itself, reached through a volatile pointer — what an unoptimised build would call.
// Inputs: (a) every x whose exact square is a TIE between two floats (x = k * 2^e, k odd, 2^24 < k^2 < 2^25);
// (b) a strided sweep of [2^-30, 8) and its negatives.
static float (*volatile libm_powf)(float, float) = powf;
long sweep_square(unsigned stride, int use_libm, long *total) {
    long bad = 0, n = 0;
    for (int k = 4097; k <= 5791; k += 2)
        for (int e = -40; e <= 20; ++e)
            for (int sgn = 0; sgn < 2; ++sgn) {
                const float x = ldexpf((float)(sgn ? -k : k), e);
                const float p1 = use_libm ? libm_powf(x, 2) : powf(x, 2), p2 = x * x; if (memcmp(&p1, &p2, 4)) bad++; n++;
            }
    for (unsigned long u = 0x30800000ul; u < 0x41000000ul; u += stride) { unsigned b = (unsigned)u; float x; memcpy(&x, &b, 4);
        const float p2 = x * x;
        float p1 = use_libm ? libm_powf(x, 2) : powf(x, 2); if (memcmp(&p1, &p2, 4)) bad++;
        p1 = use_libm ? libm_powf(-x, 2) : powf(-x, 2); if (memcmp(&p1, &p2, 4)) bad++; n += 2; }
    *total = n; return bad;
}
unsigned dm_wang(unsigned s) { return rtd::wang_hash(s); }
float dm_rand(unsigned *s) { return rtd::random_float(*s); }
unsigned char dm_tonemap(float sum, float inv) { return rtd::tonemap_u8(sum, inv); }
}
'''


@pytest.fixture(scope="module")
def dm(tmp_path_factory):
    d = tmp_path_factory.mktemp("dm")
    src = d / "dm.cpp"
    src.write_text(SRC % ROOT)
    so = d / "libdm.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    lib = C.CDLL(str(so))
    lib.sweep_exp.restype = C.c_long
    lib.sweep_exp.argtypes = [C.c_uint, C.c_uint, C.c_uint]
    lib.sweep_pow5.restype = C.c_long
    lib.sweep_pow5.argtypes = [C.c_uint, C.POINTER(C.c_long)]
    lib.sweep_identities.restype = C.c_long
    lib.sweep_identities.argtypes = [C.c_uint]
    lib.sweep_square.restype = C.c_long
    lib.sweep_square.argtypes = [C.c_uint, C.c_int, C.POINTER(C.c_long)]
    lib.dm_wang.restype = C.c_uint
    lib.dm_rand.restype = C.c_float
    lib.dm_rand.argtypes = [C.POINTER(C.c_uint)]
    lib.dm_tonemap.restype = C.c_ubyte
    lib.dm_tonemap.argtypes = [C.c_float, C.c_float]
    return lib


def test_expf_matches_host_libm(dm):
    # all non-positive floats down to -128, every 97th bit pattern (the exhaustive sweep, run once
    # while developing, found 1 mismatch in 1.12e9: x=-0x1.f8cbb2p+5, an FMA-vs-no-FMA last bit)
    bad = dm.sweep_exp(0x80000000, 0xC3000000, 97)
    assert bad <= 1
    assert dm.sweep_exp(0x00000000, 0x42B00000, 1013) <= 2     # positive side up to 88


def test_pow5_is_within_libm_noise(dm):
    total = C.c_long()
    bad = dm.sweep_pow5(53, C.byref(total))
    assert bad / total.value < 5e-4     # exhaustive: 0.0136 % of [0,1] differ from glibc powf(x,5) by 1 ulp


def test_single_op_through_double_equals_float_op(dm):
    assert dm.sweep_identities(1009) == 0


def test_square_is_what_an_optimised_build_makes_of_powf_2(dm):
    """ELLIPSE interior test (include/plane.h:41: powf(x, 2); rt_kernel.hip.inc: x * x).  With the literal exponent
    every optimising compiler expands the call to x * x — checked here on the code gcc makes of it with the oracle's
    flags (-O2, no fast-math): identical on all 103 k exact-tie inputs and on every 5th float of [2^-30, 8) and the
    negatives (111 M values).  That is the parity target: both of the reference's build files compile with -O3.
    The library routine itself (what an UNoptimised build calls) is not correctly rounded: glibc 2.35's powf(x, 2)
    differs from x * x by one ulp on ~0.07 % of the sweep and on most exact ties — measured, not asserted to be zero;
    the bound below only keeps the documented figure honest."""
    total = C.c_long()
    assert dm.sweep_square(5, 0, C.byref(total)) == 0 and total.value > 100_000_000
    bad = dm.sweep_square(5, 1, C.byref(total))
    assert bad / total.value < 2e-3


def test_rng_and_tonemap_match_oracle(dm, golden):
    import oracle_bindings as ob
    for k, v in golden["wang_hash"].items():
        assert dm.dm_wang(int(k)) == v
    s1, s2 = C.c_uint(123456789), C.c_uint(123456789)
    for _ in range(1000):
        a = dm.dm_rand(C.byref(s1))
        b = ob.lib().orc_random_float(C.byref(s2))
        assert a == b and s1.value == s2.value
    rng = np.random.default_rng(3)
    sums = np.concatenate([rng.uniform(0, 40, 3000), [0.0, -1.0, 1e9, 3.996, 3.9961]]).astype(np.float32)
    want = ob.write_color_bytes(np.stack([sums, sums, sums], 1), 4)[:, 0]
    inv = np.float32(1.0 / np.float64(np.float32(4)))
    got = np.array([dm.dm_tonemap(float(x), float(inv)) for x in sums], dtype=np.uint8)
    assert np.array_equal(got, want)
