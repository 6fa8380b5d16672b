// Microbenchmark v3: issue rates of the VALU ops that could pack MFMA result signs more cheaply than v_alignbit_b32,
// and the f16 32x32x16 filter tile (64 rays x 32 spheres = 4 MFMA + 64 VALU) with three ways of capturing the signs.
// Saturated grid, wall-clock timing, clock from s_memtime / s_memrealtime (same method as valu_rate2.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

#define OP8(TXT) asm volatile(TXT(%0) TXT(%1) TXT(%2) TXT(%3) TXT(%4) TXT(%5) TXT(%6) TXT(%7) \
    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(a), "v"(b), "v"(c) : "vcc")
#define T_FP8(D)    "v_cvt_pk_fp8_f32 " #D ", %8, %9\n"
#define T_FP8HI(D)  "v_cvt_pk_fp8_f32 " #D ", %8, %9 op_sel:[0,0,1]\n"
#define T_FP4(D)    "v_cvt_scalef32_pk_fp4_f32 " #D ", %8, %9, %10\n"
#define T_BF16(D)   "v_cvt_pk_bf16_f32 " #D ", %8, %9\n"
#define T_ANDOR(D)  "v_and_or_b32 " #D ", " #D ", %8, %9\n"
#define T_LSHLOR(D) "v_lshl_or_b32 " #D ", " #D ", 1, %9\n"
#define T_PERM(D)   "v_perm_b32 " #D ", " #D ", %8, %9\n"
#define T_MIN3(D)   "v_min3_f32 " #D ", " #D ", %8, %9\n"
#define T_BFI(D)    "v_bfi_b32 " #D ", " #D ", %8, %9\n"
#define T_CMP(D)    "v_cmp_lt_f32 vcc, " #D ", %8\n"
#define T_LSHR(D)   "v_lshrrev_b32 " #D ", 1, " #D "\n"
#define T_OR3(D)    "v_or3_b32 " #D ", " #D ", %8, %9\n"
#define T_U8(D)     "v_cvt_pk_u8_f32 " #D ", %8, 1, " #D "\n"
#define T_MAX(D)    "v_max_f32 " #D ", " #D ", %8\n"
#define T_AND(D)    "v_and_b32 " #D ", " #D ", %8\n"
#define T_PKMUL(D)  "v_pk_mul_f32 %0, %0, %8\n"
#define T_LSHLADD(D) "v_lshl_add_u32 " #D ", " #D ", 1, %9\n"
#define T_ADD3(D)   "v_add3_u32 " #D ", " #D ", %8, %9\n"
#define T_SWAP(D)   "v_permlane32_swap_b32 " #D ", %8\n"
#define T_FFBL(D)   "v_ffbl_b32 " #D ", " #D "\n"
#define T_CMPCLS(D) "v_cmp_class_f32 vcc, " #D ", %8\n"
#define T_ASHR(D)   "v_ashrrev_i32 " #D ", 31, " #D "\n"
#define T_MUL(D)    "v_mul_f32 " #D ", " #D ", %8\n"
#define T_MULLEG(D) "v_mul_legacy_f32 " #D ", " #D ", %8\n"
#define T_CVTI(D)   "v_cvt_i32_f32 " #D ", " #D "\n"
#define T_SAD(D)    "v_sad_u32 " #D ", " #D ", %8, %9\n"
#define T_ADDCO(D)  "v_add_co_u32 " #D ", vcc, " #D ", %8\n"
#define T_ADDC(D)   "v_addc_co_u32 " #D ", vcc, " #D ", %8, vcc\n"
#define T_MBCNT(D)  "v_mbcnt_lo_u32_b32 " #D ", %8, " #D "\n"
#define T_BFE(D)    "v_bfe_u32 " #D ", " #D ", 7, 5\n"
#define T_BITOP3(D) "v_bitop3_b32 " #D ", " #D ", %8, %9 bitop3:0x96\n"
#define T_ADDU(D)   "v_add_u32 " #D ", " #D ", %8\n"
#define T_ADDF(D)   "v_add_f32 " #D ", " #D ", %8\n"
#define T_OR(D)     "v_or_b32 " #D ", " #D ", %8\n"
#define T_LSHL(D)   "v_lshlrev_b32 " #D ", 1, " #D "\n"
#define T_MINF(D)   "v_min_f32 " #D ", " #D ", %8\n"
#define T_CVTF16(D) "v_cvt_f16_f32 " #D ", " #D "\n"
#define T_CNDV(D)   "v_cndmask_b32 " #D ", " #D ", %8, vcc\n"
#define T_FMAC(D)   "v_fmac_f32 " #D ", %8, %9\n"
#define T_SUBU(D)   "v_sub_u32 " #D ", " #D ", %8\n"

enum { R_FP8, R_FP8HI, R_FP4, R_BF16, R_ANDOR, R_LSHLOR, R_PERM, R_MIN3, R_BFI, R_CMP, R_LSHR, R_OR3, R_U8, R_MAX, R_AND, R_LSHLADD, R_ADD3, R_SWAP, R_FFBL, R_CMPCLS,
       R_ASHR, R_MUL, R_MULLEG, R_CVTI, R_SAD, R_ADDCO, R_ADDC, R_MBCNT, R_BFE, R_BITOP3, R_ADDU, R_ADDF, R_OR, R_LSHL, R_MINF, R_CVTF16, R_CNDV, R_FMAC, R_SUBU, R_COUNT,
       T_ALIGN = 100, T_FP8CAP, T_FP4CAP, T_MFMAONLY };
static const char* rnames[] = { "v_cvt_pk_fp8_f32", "v_cvt_pk_fp8_f32 op_sel hi", "v_cvt_scalef32_pk_fp4_f32", "v_cvt_pk_bf16_f32", "v_and_or_b32", "v_lshl_or_b32", "v_perm_b32", "v_min3_f32",
                                "v_bfi_b32", "v_cmp_lt_f32", "v_lshrrev_b32", "v_or3_b32", "v_cvt_pk_u8_f32", "v_max_f32", "v_and_b32", "v_lshl_add_u32", "v_add3_u32", "v_permlane32_swap_b32",
                                "v_ffbl_b32", "v_cmp_class_f32", "v_ashrrev_i32", "v_mul_f32", "v_mul_legacy_f32", "v_cvt_i32_f32", "v_sad_u32", "v_add_co_u32", "v_addc_co_u32", "v_mbcnt_lo", "v_bfe_u32", "v_bitop3_b32", "v_add_u32", "v_add_f32", "v_or_b32", "v_lshlrev_b32", "v_min_f32", "v_cvt_f16_f32", "v_cndmask_b32 (vcc)", "v_fmac_f32", "v_sub_u32" };

template <int MODE>
__global__ void __launch_bounds__(256) bench(float* out, int iters, unsigned long long* clk)
{
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    float    a = threadIdx.x * 1e-3f - 0.1f, b = 1.5f - a, c = 1.0f;
    h8v A0, A1, B0, B1, B2, B3;
    for (int j = 0; j < 8; ++j) { A0[j] = (_Float16)(0.5f + threadIdx.x * 0.01f + j); A1[j] = (_Float16)(0.25f * j); B0[j] = (_Float16)(0.1f * threadIdx.x); B1[j] = (_Float16)(1.0f + j);
                                  B2[j] = (_Float16)(-0.1f * j); B3[j] = (_Float16)(2.0f - j); }
    f16v keep = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == R_FP8) OP8(T_FP8);
        else if (MODE == R_FP8HI) OP8(T_FP8HI);
        else if (MODE == R_FP4) OP8(T_FP4);
        else if (MODE == R_BF16) OP8(T_BF16);
        else if (MODE == R_ANDOR) OP8(T_ANDOR);
        else if (MODE == R_LSHLOR) OP8(T_LSHLOR);
        else if (MODE == R_PERM) OP8(T_PERM);
        else if (MODE == R_MIN3) OP8(T_MIN3);
        else if (MODE == R_BFI) OP8(T_BFI);
        else if (MODE == R_CMP) OP8(T_CMP);
        else if (MODE == R_LSHR) OP8(T_LSHR);
        else if (MODE == R_OR3) OP8(T_OR3);
        else if (MODE == R_U8) OP8(T_U8);
        else if (MODE == R_MAX) OP8(T_MAX);
        else if (MODE == R_AND) OP8(T_AND);
        else if (MODE == R_LSHLADD) OP8(T_LSHLADD);
        else if (MODE == R_ADD3) OP8(T_ADD3);
        else if (MODE == R_SWAP) OP8(T_SWAP);
        else if (MODE == R_FFBL) OP8(T_FFBL);
        else if (MODE == R_CMPCLS) OP8(T_CMPCLS);
        else if (MODE == R_ASHR) OP8(T_ASHR);
        else if (MODE == R_MUL) OP8(T_MUL);
        else if (MODE == R_MULLEG) OP8(T_MULLEG);
        else if (MODE == R_CVTI) OP8(T_CVTI);
        else if (MODE == R_SAD) OP8(T_SAD);
        else if (MODE == R_ADDCO) OP8(T_ADDCO);
        else if (MODE == R_ADDC) OP8(T_ADDC);
        else if (MODE == R_MBCNT) OP8(T_MBCNT);
        else if (MODE == R_BFE) OP8(T_BFE);
        else if (MODE == R_BITOP3) OP8(T_BITOP3);
        else if (MODE == R_ADDU) OP8(T_ADDU);
        else if (MODE == R_ADDF) OP8(T_ADDF);
        else if (MODE == R_OR) OP8(T_OR);
        else if (MODE == R_LSHL) OP8(T_LSHL);
        else if (MODE == R_MINF) OP8(T_MINF);
        else if (MODE == R_CVTF16) OP8(T_CVTF16);
        else if (MODE == R_CNDV) OP8(T_CNDV);
        else if (MODE == R_FMAC) OP8(T_FMAC);
        else if (MODE == R_SUBU) OP8(T_SUBU);
        else {
            // one f16 filter tile: 64 rays x 32 spheres.  h form -> w = oo - h^2 -> c form on top -> signs.
            const f16v zero = {0};
            f16v h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B0, zero, 0, 0, 0);
            f16v h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B1, zero, 0, 0, 0);
            if (MODE != T_MFMAONLY) {
#pragma unroll
                for (int j = 0; j < 16; ++j) h0[j] = __builtin_fmaf(-h0[j], h0[j], a);
#pragma unroll
                for (int j = 0; j < 16; ++j) h1[j] = __builtin_fmaf(-h1[j], h1[j], b);
            }
            h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B2, h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B3, h1, 0, 0, 0);
            if (MODE == T_ALIGN) {
#pragma unroll
                for (int j = 0; j < 16; ++j) u0 = __builtin_amdgcn_alignbit(u0, __float_as_uint(h0[j]), 31);
#pragma unroll
                for (int j = 0; j < 16; ++j) u1 = __builtin_amdgcn_alignbit(u1, __float_as_uint(h1[j]), 31);
            } else if (MODE == T_FP8CAP) {
                // 4 values -> one dword of fp8 (sign bits 7,15,23,31); 4 dwords merged at shifts 0..3
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f16v& w = hh ? h1 : h0;
                    unsigned m = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        int d = __builtin_amdgcn_cvt_pk_fp8_f32(w[4 * k], w[4 * k + 1], 0, false);
                        d     = __builtin_amdgcn_cvt_pk_fp8_f32(w[4 * k + 2], w[4 * k + 3], d, true);
                        m     = (m >> 1) | ((unsigned)d & 0x80808080u);
                    }
                    if (hh) u1 = (u1 >> 4) | m; else u0 = (u0 >> 4) | m;
                }
            } else if (MODE == T_FP4CAP) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f16v& w = hh ? h1 : h0;
                    unsigned m = 0;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        unsigned d = 0;
                        d = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(d, w[8 * k], w[8 * k + 1], 1.0f, 0);
                        d = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(d, w[8 * k + 2], w[8 * k + 3], 1.0f, 1);
                        d = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(d, w[8 * k + 4], w[8 * k + 5], 1.0f, 2);
                        d = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(d, w[8 * k + 6], w[8 * k + 7], 1.0f, 3);
                        m = (m >> 1) | (d & 0x88888888u);
                    }
                    if (hh) u1 = (u1 >> 2) | m; else u0 = (u0 >> 2) | m;
                }
            } else {
                keep = h0; keep[0] += h1[0];
            }
            A0[0] += (_Float16)1.0f;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7) + keep[0] + keep[5];
    if (iters < 0) out[threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE> void run(int blocks_per_cu, int iters, float* out, unsigned long long* clk_d, const char* name, int ipb)
{
    const int grid = 256 * 16;
    size_t lds = 160 * 1024 / blocks_per_cu - 512;
    if (lds < 8192) lds = 8192;
    CHECK(hipFuncSetAttribute((const void*)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bench<MODE><<<grid, 256, lds>>>(out, iters / 4, clk_d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    bench<MODE><<<grid, 256, lds>>>(out, iters, clk_d);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> clk(grid * 2);
    CHECK(hipMemcpy(clk.data(), clk_d, grid * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; for (int i = 0; i < grid; ++i) { cyc += clk[2 * i]; rt += clk[2 * i + 1]; }
    double mhz = cyc / rt * 100.0;
    double per_simd = (double)grid * 4 * iters * ipb / 1024.0;
    printf("%-40s waves/SIMD=%d  ms=%8.3f  clk=%7.1f MHz  cycles per %s per SIMD = %8.3f\n", name, blocks_per_cu, ms, mhz, ipb == 1 ? "tile" : "inst", ms * 1e-3 * mhz * 1e6 / per_simd);
}

int main()
{
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 1024)); CHECK(hipMalloc(&clk, 256 * 16 * 2 * sizeof(unsigned long long)));
    const int iters = 20000;
#define RATE(M) run<M>(4, iters, out, clk, rnames[M], 8); run<M>(8, iters, out, clk, rnames[M], 8);
#ifndef ONLY_NEW
    RATE(R_FP8) RATE(R_FP8HI) RATE(R_FP4) RATE(R_BF16) RATE(R_ANDOR) RATE(R_LSHLOR) RATE(R_PERM) RATE(R_MIN3) RATE(R_BFI) RATE(R_CMP) RATE(R_LSHR) RATE(R_OR3) RATE(R_U8) RATE(R_MAX)
    RATE(R_AND) RATE(R_LSHLADD) RATE(R_ADD3) RATE(R_SWAP) RATE(R_FFBL) RATE(R_CMPCLS) RATE(R_ASHR) RATE(R_MUL) RATE(R_MULLEG) RATE(R_CVTI) RATE(R_SAD) RATE(R_ADDCO) RATE(R_ADDC) RATE(R_MBCNT) RATE(R_BFE)
#endif
    RATE(R_BITOP3) RATE(R_ADDU) RATE(R_ADDF) RATE(R_OR) RATE(R_LSHL) RATE(R_MINF) RATE(R_CVTF16) RATE(R_CNDV) RATE(R_FMAC) RATE(R_SUBU)
#ifndef ONLY_NEW
#define TILE(M, NAME) run<M>(2, iters, out, clk, NAME, 1); run<M>(3, iters, out, clk, NAME, 1); run<M>(4, iters, out, clk, NAME, 1); run<M>(8, iters, out, clk, NAME, 1);
    TILE(T_MFMAONLY, "f16 tile 64x32: 4 mfma only")
    TILE(T_ALIGN, "f16 tile 64x32: 4 mfma + 32 fma + 32 alignbit")
    TILE(T_FP8CAP, "f16 tile 64x32: fp8 sign capture")
    TILE(T_FP4CAP, "f16 tile 64x32: fp4 sign capture")
#endif
    return 0;
}
