// Microbenchmark: issue cost of the f16 MFMA shapes on gfx950, alone and inside the filter-tile pattern
// (h form -> w = oo - h^2 -> c form -> sign capture), to pick the shape for a two-way f16 split (K = 16 per form).
// Saturated grid, wall-clock timing (same method as valu_rate2.hip).  Every line is "per 64 rays x 32 spheres".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
enum { M_16x16x16_ONLY, M_16x16x32_ONLY, M_32x32x8_ONLY, M_32x32x16_ONLY, M_4x4x4_ONLY, M_16x16x16_TILE, M_16x16x32_TILE, M_32x32x8_TILE, M_VALU_ONLY };

template <int MODE>
__global__ void __launch_bounds__(256) bench(float* out, int iters, unsigned long long* clk)
{
    h4v a4, b4[4], c4[4];
    h8v a8, b8[4], c8[4];
    for (int j = 0; j < 4; ++j) a4[j] = (_Float16)(0.5f + threadIdx.x * 0.01f + j);
    for (int j = 0; j < 8; ++j) a8[j] = (_Float16)(0.5f + threadIdx.x * 0.01f + j);
    for (int r = 0; r < 4; ++r) {
        for (int j = 0; j < 4; ++j) { b4[r][j] = (_Float16)(0.1f * threadIdx.x + r); c4[r][j] = (_Float16)(1.0f + j - r); }
        for (int j = 0; j < 8; ++j) { b8[r][j] = (_Float16)(0.1f * threadIdx.x + r); c8[r][j] = (_Float16)(1.0f + j - r); }
    }
    float oo[4] = { threadIdx.x * 1e-3f, 0.5f, 0.25f, 0.125f };
    unsigned mk[4] = { 0, 0, 0, 0 };
    f4v  keep4 = {0}, keepb = {0};
    f16v keep16 = {0}, keep16b = {0};
    const f4v  z4 = {0};
    const f16v z16 = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_16x16x16_ONLY) {            // 16 MFMA = 2 tiles x 4 ray groups x 2 forms
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { keep4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4[r], keep4, 0, 0, 0); keepb = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, c4[r], keepb, 0, 0, 0); }
        } else if (MODE == M_16x16x32_ONLY) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { keep4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8[r], keep4, 0, 0, 0); keepb = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, c8[r], keepb, 0, 0, 0); }
        } else if (MODE == M_32x32x8_ONLY) {      // K = 8 only: two per form -> 8 MFMA
#pragma unroll
            for (int r = 0; r < 2; ++r) { keep16 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4[r], keep16, 0, 0, 0); keep16b = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4[r + 2], keep16b, 0, 0, 0);
                                          keep16 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, c4[r], keep16, 0, 0, 0); keep16b = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, c4[r + 2], keep16b, 0, 0, 0); }
        } else if (MODE == M_32x32x16_ONLY) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { keep16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8[r], keep16, 0, 0, 0); keep16b = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, c8[r], keep16b, 0, 0, 0); }
        } else if (MODE == M_4x4x4_ONLY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) keep4 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4[r & 3], keep4, 0, 0, 0);
        } else if (MODE == M_16x16x16_TILE || MODE == M_16x16x32_TILE) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int rp = 0; rp < 4; rp += 2) {
                    f4v h0, h1;
                    if (MODE == M_16x16x16_TILE) { h0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4[rp], z4, 0, 0, 0); h1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4[rp + 1], z4, 0, 0, 0); }
                    else                         { h0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8[rp], z4, 0, 0, 0); h1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8[rp + 1], z4, 0, 0, 0); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) h0[j] = __builtin_fmaf(-h0[j], h0[j], oo[rp]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) h1[j] = __builtin_fmaf(-h1[j], h1[j], oo[rp + 1]);
                    if (MODE == M_16x16x16_TILE) { h0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, c4[rp], h0, 0, 0, 0); h1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, c4[rp + 1], h1, 0, 0, 0); }
                    else                         { h0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, c8[rp], h0, 0, 0, 0); h1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, c8[rp + 1], h1, 0, 0, 0); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) mk[rp] = __builtin_amdgcn_alignbit(mk[rp], __float_as_uint(h0[j]), 31);
#pragma unroll
                    for (int j = 0; j < 4; ++j) mk[rp + 1] = __builtin_amdgcn_alignbit(mk[rp + 1], __float_as_uint(h1[j]), 31);
                }
                { uint2 w = __builtin_bit_cast(uint2, a4); w.x ^= (unsigned)it; a4 = __builtin_bit_cast(h4v, w); uint4 w8 = __builtin_bit_cast(uint4, a8); w8.x ^= (unsigned)it; a8 = __builtin_bit_cast(h8v, w8); }
            }
        } else if (MODE == M_32x32x8_TILE) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f16v h = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4[r], z16, 0, 0, 0);
                h = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4[r + 2], h, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 16; ++j) h[j] = __builtin_fmaf(-h[j], h[j], oo[r]);
                h = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, c4[r], h, 0, 0, 0);
                h = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, c4[r + 2], h, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 16; ++j) mk[r] = __builtin_amdgcn_alignbit(mk[r], __float_as_uint(h[j]), 31);
            }
            { uint2 w = __builtin_bit_cast(uint2, a4); w.x ^= (unsigned)it; a4 = __builtin_bit_cast(h4v, w); }
        } else if (MODE == M_VALU_ONLY) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int j = 0; j < 16; ++j) keep16[j] = __builtin_fmaf(-keep16[j], keep16[j], oo[r]);
#pragma unroll
                for (int j = 0; j < 16; ++j) mk[r] = __builtin_amdgcn_alignbit(mk[r], __float_as_uint(keep16[j]), 31);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)(mk[0] + mk[1] + mk[2] + mk[3]) + keep4[0] + keep4[3] + keep16[0] + keep16[7] + keepb[1] + keep16b[2];
    if (iters < 0) out[threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE> void run(int blocks_per_cu, int iters, float* out, unsigned long long* clk_d, const char* name)
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
    printf("%-52s waves/SIMD=%d  ms=%8.3f  clk=%7.1f MHz  cycles per 64x32 block per SIMD = %8.2f\n", name, blocks_per_cu, ms, mhz, ms * 1e-3 * mhz * 1e6 / per_simd);
}

int main()
{
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 1024)); CHECK(hipMalloc(&clk, 256 * 16 * 2 * sizeof(unsigned long long)));
    const int iters = 10000;
#define RUN(M, NAME) run<M>(2, iters, out, clk, NAME); run<M>(3, iters, out, clk, NAME); run<M>(4, iters, out, clk, NAME); run<M>(8, iters, out, clk, NAME);
    RUN(M_16x16x16_ONLY, "16 x mfma_f32_16x16x16_f16")
    RUN(M_16x16x32_ONLY, "16 x mfma_f32_16x16x32_f16")
    RUN(M_32x32x8_ONLY,  "8 x mfma_f32_32x32x8_f16")
    RUN(M_32x32x16_ONLY, "4 x mfma_f32_32x32x16_f16")
    RUN(M_4x4x4_ONLY,    "16 x mfma_f32_4x4x4_16B_f16 (not a full block)")
    RUN(M_VALU_ONLY,     "64 valu (32 fma + 32 alignbit)")
    RUN(M_16x16x16_TILE, "tile: 16 x 16x16x16_f16 + 64 valu")
    RUN(M_16x16x32_TILE, "tile: 16 x 16x16x32_f16 + 64 valu")
    RUN(M_32x32x8_TILE,  "tile: 8 x 32x32x8_f16 + 64 valu")
    return 0;
}
