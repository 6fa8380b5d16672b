// Microbenchmark: can v_cvt_scalef32_2xpk16_fp6_f32 (32 floats in two 16-register operands -> 32 FP6 values in 6 dwords)
// capture the 32 signs of a filter tile in ONE instruction?  Part 1 probes its semantics (which output bit carries the
// sign of which input, what happens to tiny / zero / infinite / NaN inputs, what `scale` does); part 2 measures its issue
// cost, alone, as a 32-sign pack next to 32 x v_alignbit_b32, and inside a filter tile (4 MFMA + the pack).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float    f16v __attribute__((ext_vector_type(16)));
typedef unsigned u6v __attribute__((ext_vector_type(6)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// ---- part 1: semantics ---------------------------------------------------------------------------------------------
__global__ void probe(const float* in, const float* scales, int cases, unsigned* out)
{
    for (int c = 0; c < cases; ++c) {
        f16v a, b;
        for (int j = 0; j < 16; ++j) { a[j] = in[c * 32 + j]; b[j] = in[c * 32 + 16 + j]; }
        const u6v r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scales[c]);
        if (threadIdx.x == 0) for (int k = 0; k < 6; ++k) out[c * 6 + k] = r[k];
    }
}

// ---- part 2: cost -----------------------------------------------------------------------------------------------------
enum { M_CVT, M_PACK_CVT, M_PACK_ALIGNBIT, M_TILE_CVT, M_TILE_ALIGNBIT, M_MFMA_ONLY };
static const char* names[] = { "v_cvt_scalef32_2xpk16_fp6_f32 alone", "pack 32 signs: cvt + 7 bitwise", "pack 32 signs: 32 x v_alignbit_b32",
                               "tile: 4 mfma + cvt pack", "tile: 4 mfma + 32 alignbit", "tile: 4 mfma only" };

__device__ __forceinline__ unsigned pack_cvt(const f16v& h0, const f16v& h1)
{
    const u6v r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(h0, h1, 1.0f);
    // sign bits sit at the top of every 6-bit field: bits 5, 11, 17, 23, 29 | 3, 9, 15, 21, 27 | 1, 7, 13, 19, 25, 31 of
    // dwords (0, 1, 2) and again of (3, 4, 5): merge each triple by mask (positions are disjoint), shift one, merge
    const unsigned m0 = 0x20820820u, m2 = 0x82082082u;
    unsigned t, u;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(t) : "v"(r[0]), "v"(r[1]), "v"(m0));     // (a & m) | (b & ~m)
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(t) : "v"(r[2]), "v"(t), "v"(m2));
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(u) : "v"(r[3]), "v"(r[4]), "v"(m0));
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(u) : "v"(r[5]), "v"(u), "v"(m2));
    u >>= 1;
    unsigned w;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(w) : "v"(t), "v"(u), "v"(0xAAAAAAAAu));
    return w;
}

template <int MODE>
__global__ void __launch_bounds__(256) bench(unsigned* out, int iters, unsigned long long* clk)
{
    f16v h0, h1;
    for (int j = 0; j < 16; ++j) { h0[j] = (float)((threadIdx.x * 2654435761u) ^ (j * 40503u)) - 2.0e9f; h1[j] = h0[j] * -0.5f; }
    h8v A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (_Float16)(float)((threadIdx.x + j) & 7); B[j] = (_Float16)(float)((threadIdx.x * 3 + j) & 7) - (_Float16)3.0f; }
    unsigned acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_TILE_CVT || MODE == M_TILE_ALIGNBIT || MODE == M_MFMA_ONLY) {
            const f16v zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, zero, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(B, A, zero, 0, 0, 0);
            h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(B, B, h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, A, h1, 0, 0, 0);
        }
        if (MODE == M_CVT) {
            const u6v r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(h0, h1, 1.0f);
            acc ^= r[0] ^ r[5];
            h0[0] = __uint_as_float(__float_as_uint(h0[0]) ^ (acc & 0x80000000u));
        } else if (MODE == M_PACK_CVT || MODE == M_TILE_CVT) {
            const unsigned w = pack_cvt(h0, h1);
            acc ^= w;
            if (MODE == M_PACK_CVT) h0[0] = __uint_as_float(__float_as_uint(h0[0]) ^ (acc & 0x80000000u));
            else A[0] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, A[0]) ^ (unsigned short)(acc & 0x8000u)));
        } else if (MODE == M_PACK_ALIGNBIT || MODE == M_TILE_ALIGNBIT) {
            unsigned r = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) r = __builtin_amdgcn_alignbit(r, __float_as_uint(h0[j]), 31);
#pragma unroll
            for (int j = 0; j < 16; ++j) r = __builtin_amdgcn_alignbit(r, __float_as_uint(h1[j]), 31);
            acc ^= r;
            if (MODE == M_PACK_ALIGNBIT) h0[0] = __uint_as_float(__float_as_uint(h0[0]) ^ (acc & 0x80000000u));
            else A[0] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, A[0]) ^ (unsigned short)(acc & 0x8000u)));
        } else {
            acc ^= __float_as_uint(h0[3]) ^ __float_as_uint(h1[7]);
            A[0] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, A[0]) ^ (unsigned short)(acc & 0x8000u)));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (iters < 0) out[threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE> void run(int blocks_per_cu, int iters, unsigned* out, unsigned long long* clk_d)
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
    double per_simd = (double)grid * 4 * iters / 1024.0;
    printf("%-40s waves/SIMD=%d  ms=%8.3f  clk=%7.1f MHz  cycles per loop body per SIMD = %8.3f\n", names[MODE], blocks_per_cu, ms, mhz, ms * 1e-3 * mhz * 1e6 / per_simd);
    fflush(stdout);
}

int main()
{
    // ---- part 1
    std::vector<float> in; std::vector<float> scales; std::vector<const char*> label;
    auto add_case = [&](const char* name, float scale, int idx, float x) {
        for (int j = 0; j < 32; ++j) in.push_back(j == idx ? x : 1.0f);
        scales.push_back(scale); label.push_back(name);
    };
    add_case("all +1", 1.0f, -1, 0.0f);
    for (int k = 0; k < 32; ++k) add_case("-1 at k", 1.0f, k, -1.0f);
    const float ninf = -__builtin_inff();
    float nnan, pnan; { unsigned u = 0xFFC00000u; memcpy(&nnan, &u, 4); u = 0x7FC00000u; memcpy(&pnan, &u, 4); }
    float ndenorm; { unsigned u = 0x80000001u; memcpy(&ndenorm, &u, 4); }
    struct { const char* n; float x; } specials[] = { { "-0.0", -0.0f }, { "+0.0", 0.0f }, { "-1e-30", -1e-30f }, { "+1e-30", 1e-30f }, { "-denormal", ndenorm }, { "-0.06", -0.06f }, { "-0.07", -0.07f },
                                                       { "-3e38", -3e38f }, { "-inf", ninf }, { "-nan", nnan }, { "+nan", pnan }, { "-7.5", -7.5f }, { "-100", -100.0f } };
    const int first_special = (int)scales.size();
    for (auto& s : specials) add_case(s.n, 1.0f, 0, s.x);
    const int first_scaled = (int)scales.size();
    const float sc[] = { 0.0f, 1e-30f, 7.8886e-31f /* 2^-100 */, 0.125f, 8.0f, 1.2677e30f /* 2^100 */ };
    for (float s : sc) { add_case("-1 under scale", s, 0, -1.0f); add_case("-1e-30 under scale", s, 0, -1e-30f); add_case("-1e30 under scale", s, 0, -1e30f); }
    const int cases = (int)scales.size();
    float *in_d, *sc_d; unsigned* o_d;
    CHECK(hipMalloc(&in_d, in.size() * 4)); CHECK(hipMalloc(&sc_d, cases * 4)); CHECK(hipMalloc(&o_d, cases * 6 * 4));
    CHECK(hipMemcpy(in_d, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(sc_d, scales.data(), cases * 4, hipMemcpyHostToDevice));
    probe<<<1, 64>>>(in_d, sc_d, cases, o_d);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> o(cases * 6);
    CHECK(hipMemcpy(o.data(), o_d, cases * 6 * 4, hipMemcpyDeviceToHost));
    printf("all +1      : %08x %08x %08x %08x %08x %08x\n", o[0], o[1], o[2], o[3], o[4], o[5]);
    printf("input k -> output bits that change when it becomes -1 (dword:bit):\n");
    for (int k = 0; k < 32; ++k) {
        printf("  k=%2d:", k);
        for (int d = 0; d < 6; ++d) { unsigned x = o[(1 + k) * 6 + d] ^ o[d]; for (int b = 0; b < 32; ++b) if (x >> b & 1) printf(" %d:%d", d, b); }
        printf("\n");
    }
    for (int c = first_special; c < cases; ++c) {
        const unsigned field0 = o[c * 6] & 63u;
        printf("%-20s scale=%-12g input0=%-12g -> field 0 = %02x (sign %u, rest %02x)%s\n", label[c], scales[c], in[c * 32], field0, field0 >> 5, field0 & 31u, c == first_scaled ? "   [scaled cases]" : "");
    }
    // ---- part 2
    unsigned* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 1024)); CHECK(hipMalloc(&clk, 256 * 16 * 2 * sizeof(unsigned long long)));
    const int iters = 20000;
#define RUN(M) run<M>(2, iters, out, clk); run<M>(4, iters, out, clk); run<M>(8, iters, out, clk);
    RUN(M_CVT) RUN(M_PACK_CVT) RUN(M_PACK_ALIGNBIT) RUN(M_MFMA_ONLY) RUN(M_TILE_CVT) RUN(M_TILE_ALIGNBIT)
    return 0;
}
