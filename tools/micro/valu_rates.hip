// Microbenchmark: sustained issue cost (cycles per wave-instruction per SIMD) of the VALU instruction kinds the trace
// kernel is made of, at 6 waves per SIMD (768-thread workgroups x 2 per CU) and at 1 wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rates.bin tools/micro/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define BODY(INSTR) \
    for (int i = 0; i < iters; ++i) { \
        asm volatile(REP8(INSTR(%0) INSTR(%1) INSTR(%2) INSTR(%3) INSTR(%4) INSTR(%5) INSTR(%6) INSTR(%7)) \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(x), "v"(y), "s"(k), "s"(mask) : "vcc", "scc", "s20", "s21"); }
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define I_ADD(r) "v_add_f32 " #r ", " #r ", %8\n"
#define I_MIN(r) "v_min_f32 " #r ", " #r ", %8\n"
#define I_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9\n"
#define I_XOR(r) "v_xor_b32 " #r ", " #r ", %8\n"
#define I_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 3, %8\n"
#define I_LSHR(r) "v_lshrrev_b32 " #r ", 4, " #r "\n"
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %10\n"
#define I_MAD16(r) "v_mad_u32_u16 " #r ", " #r ", %10, %8\n"
#define I_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", %8\n"
#define I_CNDMASK(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, %11\n"
#define I_CNDVCC(r) "v_cndmask_b32_e32 " #r ", " #r ", %8, vcc\n"
#define I_SUB(r) "v_sub_f32 " #r ", " #r ", %8\n"
#define I_MUL(r) "v_mul_f32 " #r ", " #r ", %8\n"
#define I_MAX(r) "v_max_f32 " #r ", " #r ", %8\n"
#define I_MIN3(r) "v_min3_f32 " #r ", " #r ", %8, %9\n"
#define I_MED3(r) "v_med3_f32 " #r ", " #r ", %8, %9\n"
#define I_AND(r) "v_and_b32 " #r ", " #r ", %8\n"
#define I_OR(r) "v_or_b32 " #r ", " #r ", %8\n"
#define I_ANDOR(r) "v_and_or_b32 " #r ", " #r ", %8, %9\n"
#define I_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 5\n"
#define I_LSHL(r) "v_lshlrev_b32 " #r ", 3, " #r "\n"
#define I_CMPI(r) "v_cmp_lt_i32 vcc, " #r ", %8\n"
#define I_CMPE64(r) "v_cmp_gt_f32_e64 s[20:21], " #r ", %8\n"
#define I_ADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9\n"
#define I_XAD(r) "v_xad_u32 " #r ", " #r ", %8, %9\n"
#define I_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", %8\n"
#define I_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9\n"
#define I_CVTU(r) "v_cvt_u32_f32 " #r ", " #r "\n"
#define I_FMAMIX(r) "v_fmac_f32 " #r ", %8, %9\n"
#define I_RSQ(r) "v_rsq_f32 " #r ", " #r "\n"
#define I_DIVSCALE(r) "v_div_scale_f32 " #r ", vcc, " #r ", %8, " #r "\n"
#define I_DIVFMAS(r) "v_div_fmas_f32 " #r ", " #r ", %8, %9\n"
#define I_DIVFIXUP(r) "v_div_fixup_f32 " #r ", " #r ", %8, %9\n"
#define I_CMPCLASS(r) "v_cmp_class_f32 vcc, " #r ", %8\n"
#define I_FREXP(r) "v_frexp_mant_f32 " #r ", " #r "\n"
#define I_LDEXP(r) "v_ldexp_f32 " #r ", " #r ", %8\n"
#define I_FLOOR(r) "v_floor_f32 " #r ", " #r "\n"
#define I_BFI(r) "v_bfi_b32 " #r ", " #r ", %8, %9\n"
#define I_ASHR(r) "v_ashrrev_i32 " #r ", 3, " #r "\n"
#define I_SUBU(r) "v_sub_u32 " #r ", " #r ", %8\n"
#define I_LSHLADD64(r) "v_lshl_add_u64 %0, %0, 4, %0\n"
#define I_FMAMIXH(r) "v_fma_mix_f32 " #r ", " #r ", %8, %9 op_sel_hi:[1,0,0]\n"
#define I_CVTH(r) "v_cvt_f32_f16 " #r ", " #r "\n"
#define I_SNOP(r) "s_nop 0\n"
#define I_SALU(r) "s_and_b64 s[20:21], s[20:21], exec\n"
#define I_MBCNT(r) "v_mbcnt_lo_u32_b32 " #r ", %10, " #r "\n"
#define I_READLANE(r) "v_readfirstlane_b32 s20, " #r "\n"
#define I_MOV(r) "v_mov_b32 " #r ", %8\n"
#define I_CVT(r) "v_cvt_f32_u32 " #r ", " #r "\n"
#define I_RCP(r) "v_rcp_f32 " #r ", " #r "\n"
#define I_SQRT(r) "v_sqrt_f32 " #r ", " #r "\n"
#define I_CMP(r) "v_cmp_gt_f32 vcc, " #r ", %8\n"
#define I_ADDU(r) "v_add_u32 " #r ", " #r ", %8\n"
#define I_XORSDWA(r) "v_xor_b32_sdwa " #r ", " #r ", " #r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
template <int kMode> __global__ void __launch_bounds__(768, 2) k(uint32_t *out, int iters, uint32_t k) {
    uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    uint32_t x = 0x3f800001u, y = 0x3a000000u;
    const uint64_t mask = 0x5555aaaa0f0f3333ull ^ k;
    asm volatile("v_cmp_gt_u32 vcc, 5, %0" :: "v"(r0) : "vcc");
    if (kMode == 0) BODY(I_FMA) else if (kMode == 1) BODY(I_ADD) else if (kMode == 2) BODY(I_MIN) else if (kMode == 3) BODY(I_MAX3)
    else if (kMode == 4) BODY(I_XOR) else if (kMode == 5) BODY(I_LSHLADD) else if (kMode == 6) BODY(I_LSHR) else if (kMode == 7) BODY(I_MULLO)
    else if (kMode == 8) BODY(I_MAD16) else if (kMode == 9) BODY(I_MUL24) else if (kMode == 10) BODY(I_CNDMASK) else if (kMode == 11) BODY(I_MOV)
    else if (kMode == 12) BODY(I_CVT) else if (kMode == 13) BODY(I_RCP) else if (kMode == 14) BODY(I_SQRT) else if (kMode == 15) BODY(I_CMP)
    else if (kMode == 16) BODY(I_ADDU) else if (kMode == 17) BODY(I_XORSDWA) else if (kMode == 18) BODY(I_CNDVCC) else if (kMode == 19) BODY(I_SUB)
    else if (kMode == 20) BODY(I_MUL) else if (kMode == 21) BODY(I_MAX) else if (kMode == 22) BODY(I_MIN3) else if (kMode == 23) BODY(I_MED3)
    else if (kMode == 24) BODY(I_AND) else if (kMode == 25) BODY(I_OR) else if (kMode == 26) BODY(I_ANDOR) else if (kMode == 27) BODY(I_BFE)
    else if (kMode == 28) BODY(I_LSHL) else if (kMode == 29) BODY(I_CMPI) else if (kMode == 30) BODY(I_CMPE64) else if (kMode == 31) BODY(I_ADD3)
    else if (kMode == 32) BODY(I_XAD) else if (kMode == 33) BODY(I_MULHI) else if (kMode == 34) BODY(I_MAD24) else if (kMode == 35) BODY(I_CVTU)
    else if (kMode == 36) BODY(I_FMAMIX) else if (kMode == 37) BODY(I_RSQ) else if (kMode == 38) BODY(I_SNOP) else if (kMode == 39) BODY(I_SALU)
    else if (kMode == 40) BODY(I_MBCNT) else if (kMode == 41) BODY(I_READLANE) else if (kMode == 42) BODY(I_DIVSCALE) else if (kMode == 43) BODY(I_DIVFMAS)
    else if (kMode == 44) BODY(I_DIVFIXUP) else if (kMode == 45) BODY(I_CMPCLASS) else if (kMode == 46) BODY(I_FREXP) else if (kMode == 47) BODY(I_LDEXP)
    else if (kMode == 52) BODY(I_FMAMIXH) else if (kMode == 53) BODY(I_CVTH)
    else if (kMode == 48) BODY(I_FLOOR) else if (kMode == 49) BODY(I_BFI) else if (kMode == 50) BODY(I_ASHR) else if (kMode == 51) BODY(I_SUBU)
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

#define BODY64(INSTR) \
    for (int i = 0; i < iters; ++i) { \
        asm volatile(REP8(INSTR(%0) INSTR(%1) INSTR(%2) INSTR(%3)) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(xd)); }
#define D_PKADD(r) "v_pk_add_f32 " #r ", " #r ", %8\n"
#define D_PKMUL(r) "v_pk_mul_f32 " #r ", " #r ", %8\n"
#define D_PKFMA(r) "v_pk_fma_f32 " #r ", " #r ", %8, %8\n"
#define D_RCP(r) "v_rcp_f64 " #r ", " #r "\n"
#define D_FMA(r) "v_fma_f64 " #r ", " #r ", %8, %8\n"
#define D_ADD(r) "v_add_f64 " #r ", " #r ", %8\n"
#define D_MUL(r) "v_mul_f64 " #r ", " #r ", %8\n"
#define D_DIVSCALE(r) "v_div_scale_f64 " #r ", vcc, " #r ", %8, " #r "\n"
#define D_DIVFMAS(r) "v_div_fmas_f64 " #r ", " #r ", %8, %8\n"
#define D_DIVFIXUP(r) "v_div_fixup_f64 " #r ", " #r ", %8, %8\n"
#define D_CVTDF(r) "v_cvt_f64_f32 %0, %4\n"
#define D_CVTFD(r) "v_cvt_f32_f64 %4, %0\n"
template <int kMode> __global__ void __launch_bounds__(768, 2) k64(uint32_t *out, int iters, uint32_t k) {
    double d0 = threadIdx.x + 1.0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, xd = 1.0000001;
    float f0 = 1.5f, f1 = 2.5f, f2 = 3.5f, f3 = 4.5f;
    if (kMode == 0) BODY64(D_PKADD) else if (kMode == 1) BODY64(D_PKMUL) else if (kMode == 2) BODY64(D_PKFMA) else if (kMode == 3) BODY64(D_RCP)
    else if (kMode == 4) BODY64(D_FMA) else if (kMode == 5) BODY64(D_ADD) else if (kMode == 6) BODY64(D_MUL) else if (kMode == 7) BODY64(D_DIVSCALE)
    else if (kMode == 8) BODY64(D_DIVFMAS) else if (kMode == 9) BODY64(D_DIVFIXUP) else if (kMode == 10) BODY64(D_CVTDF) else if (kMode == 11) BODY64(D_CVTFD)
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 + d1 + d2 + d3) + (uint32_t)(f0 + f1 + f2 + f3);
}
template <int kMode> static void run64(const char *what, uint32_t *o) {
    const int iters = 4000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
        (void)hipEventRecord(s);
        hipLaunchKernelGGL(k64<kMode>, dim3(512), dim3(768), 0, 0, o, iters, 0x27d4eb2du);
        (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    printf("%-22s 6 wave(s)/SIMD: %7.3f ms = %5.2f cycles per instruction per SIMD at 2.4 GHz\n", what, best, best * 1e-3 * 2.4e9 / ((double)iters * 32 * 6)); fflush(stdout);
}
template <int kMode> static void run(const char *what, uint32_t *o) {
    const int iters = 4000;
    for (int waves = 6; waves >= 1; waves -= 5) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
            (void)hipEventRecord(s);
            if (waves == 6) hipLaunchKernelGGL(k<kMode>, dim3(512), dim3(768), 0, 0, o, iters, 0x27d4eb2du);
            else hipLaunchKernelGGL(k<kMode>, dim3(256), dim3(256), 0, 0, o, iters, 0x27d4eb2du);
            (void)hipEventRecord(e); (void)hipEventSynchronize(e);
            float ms; (void)hipEventElapsedTime(&ms, s, e);
            if (ms < best) best = ms;
        }
        printf("%-22s %d wave(s)/SIMD: %7.3f ms = %5.2f cycles per instruction per SIMD at 2.4 GHz\n", what, waves, best, best * 1e-3 * 2.4e9 / ((double)iters * 64 * waves)); fflush(stdout);
    }
}
__global__ void __launch_bounds__(768, 2) kdiv(float *out, int iters, int mode) {
    float a = threadIdx.x + 1.5f, b = a * 0.37f + 2.0f, c = b + 3.25f, d = c * 1.5f;
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) { a = 1.0f / a + 0.75f; b = 1.0f / b + 0.75f; c = 1.0f / c + 0.75f; d = 1.0f / d + 0.75f; }
        else if (mode == 1) { a = sqrtf(a) + 1.5f; b = sqrtf(b) + 1.5f; c = sqrtf(c) + 1.5f; d = sqrtf(d) + 1.5f; }
        else if (mode == 2) { a = (float)(1.0 / (double)a) + 0.75f; b = (float)(1.0 / (double)b) + 0.75f; c = (float)(1.0 / (double)c) + 0.75f; d = (float)(1.0 / (double)d) + 0.75f; }
        else { a = __builtin_amdgcn_rcpf(a) + 0.75f; b = __builtin_amdgcn_rcpf(b) + 0.75f; c = __builtin_amdgcn_rcpf(c) + 0.75f; d = __builtin_amdgcn_rcpf(d) + 0.75f; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
static void rundiv(const char *what, uint32_t *o, int mode) {
    const int iters = 20000;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
        (void)hipEventRecord(s);
        hipLaunchKernelGGL(kdiv, dim3(512), dim3(768), 0, 0, (float *)o, iters, mode);
        (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    printf("%-44s %7.3f ms = %6.1f cycles per operation (+1 add) per SIMD at 2.4 GHz\n", what, best, best * 1e-3 * 2.4e9 / ((double)iters * 4 * 6)); fflush(stdout);
}
int main() {
    uint32_t *o; (void)hipMalloc(&o, 512 * 768 * 4);
    run<0>("v_fma_f32", o); run<1>("v_add_f32", o); run<2>("v_min_f32", o); run<3>("v_max3_f32", o); run<4>("v_xor_b32", o);
    run<17>("v_xor_b32_sdwa", o); run<5>("v_lshl_add_u32", o); run<6>("v_lshrrev_b32", o); run<16>("v_add_u32", o); run<7>("v_mul_lo_u32", o);
    run<8>("v_mad_u32_u16", o); run<9>("v_mul_u32_u24", o); run<10>("v_cndmask_b32", o); run<11>("v_mov_b32", o); run<12>("v_cvt_f32_u32", o);
    run<15>("v_cmp_gt_f32", o); run<13>("v_rcp_f32", o); run<14>("v_sqrt_f32", o);
    run<18>("v_cndmask_b32 (vcc)", o); run<19>("v_sub_f32", o); run<20>("v_mul_f32", o); run<21>("v_max_f32", o); run<22>("v_min3_f32", o); run<23>("v_med3_f32", o);
    run<24>("v_and_b32", o); run<25>("v_or_b32", o); run<26>("v_and_or_b32", o); run<27>("v_bfe_u32", o); run<28>("v_lshlrev_b32", o); run<29>("v_cmp_lt_i32", o);
    run<30>("v_cmp_gt_f32_e64", o); run<31>("v_add3_u32", o); run<32>("v_xad_u32", o); run<33>("v_mul_hi_u32", o); run<34>("v_mad_u32_u24", o); run<35>("v_cvt_u32_f32", o);
    run<36>("v_fmac_f32", o); run<37>("v_rsq_f32", o); run<38>("s_nop 0", o); run<39>("s_and_b64", o); run<40>("v_mbcnt_lo", o); run<41>("v_readfirstlane", o);
    run<42>("v_div_scale_f32", o); run<43>("v_div_fmas_f32", o); run<44>("v_div_fixup_f32", o); run<45>("v_cmp_class_f32", o); run<46>("v_frexp_mant_f32", o);
    run<47>("v_ldexp_f32", o); run<48>("v_floor_f32", o); run<49>("v_bfi_b32", o); run<50>("v_ashrrev_i32", o); run<51>("v_sub_u32", o);
    run<52>("v_fma_mix_f32 (f16 src)", o); run<53>("v_cvt_f32_f16", o);
    run64<0>("v_pk_add_f32", o); run64<1>("v_pk_mul_f32", o); run64<2>("v_pk_fma_f32", o); run64<3>("v_rcp_f64", o); run64<4>("v_fma_f64", o); run64<5>("v_add_f64", o);
    run64<6>("v_mul_f64", o); run64<7>("v_div_scale_f64", o); run64<8>("v_div_fmas_f64", o); run64<9>("v_div_fixup_f64", o); run64<10>("v_cvt_f64_f32", o); run64<11>("v_cvt_f32_f64", o);
    rundiv("1.0f / x (IEEE, -fhip-fp32-correctly-rounded-divide-sqrt)", o, 0); rundiv("sqrtf(x) (IEEE)", o, 1); rundiv("(float)(1.0 / (double)x)", o, 2); rundiv("v_rcp_f32", o, 3);
    return 0;
}
