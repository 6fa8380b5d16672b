// Microbenchmark: do bf16 MFMAs (matrix cores) overlap with VALU work of other waves on the same SIMD?
// Tile = what the split-bf16 candidate filter does for 32 rays x 32 spheres:
//   H = 2 x mfma_32x32x16_bf16 ; W = fma(-h,h,oo) x16 ; C = 2 x mfma on top of W ; 16 x alignbit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
enum { M_BF16_ONLY, M_TILE32, M_TILE32_NOVALU, M_VALU_ONLY, M_TILE16, M_COUNT };
static const char* names[] = { "4 x mfma_32x32x16_bf16 (indep acc)", "bf16 tile32: 4 mfma + 32 valu", "bf16 tile32 chain, no valu", "32 valu only (16 fma + 16 alignbit)", "bf16 16x16x32 x8 + 32 valu (64 rays x 16 sph)" };

template <int MODE>
__global__ void __launch_bounds__(256) bench(float* out, int iters, unsigned long long* clk)
{
    s8v a0, a1, b0, b1;
    for (int j = 0; j < 8; ++j) { a0[j] = (short)(0x3f80 + threadIdx.x + j); a1[j] = (short)(0x3f00 + j); b0[j] = (short)(0x3e80 + threadIdx.x); b1[j] = (short)(0x3f81 + j); }
    f16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    float oo = threadIdx.x * 1e-3f;
    unsigned u0 = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_BF16_ONLY) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc3, 0, 0, 0);
        } else if (MODE == M_TILE32 || MODE == M_TILE32_NOVALU) {
            f16v h = {0};
            h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, h, 0, 0, 0);
            h = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, h, 0, 0, 0);
            f16v w = h;
            if (MODE == M_TILE32) {
#pragma unroll
                for (int j = 0; j < 16; ++j) w[j] = __builtin_fmaf(-h[j], h[j], oo);
            }
            w = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, w, 0, 0, 0);
            w = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, w, 0, 0, 0);
            if (MODE == M_TILE32) {
#pragma unroll
                for (int j = 0; j < 16; ++j) u0 = __builtin_amdgcn_alignbit(u0, __float_as_uint(w[j]), 31);
            } else {
                acc0 = w;
            }
            a0[0] ^= (short)(it & 3);
        } else if (MODE == M_VALU_ONLY) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc0[j] = __builtin_fmaf(-acc0[j], acc1[j], oo);
#pragma unroll
            for (int j = 0; j < 16; ++j) u0 = __builtin_amdgcn_alignbit(u0, __float_as_uint(acc0[j]), 31);
        } else if (MODE == M_TILE16) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f4v h = {0};
                h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, g & 1 ? b0 : b1, h, 0, 0, 0);
                f4v w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = __builtin_fmaf(-h[j], h[j], oo);
                w = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, g & 2 ? b0 : b1, w, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) u0 = __builtin_amdgcn_alignbit(u0, __float_as_uint(w[j]), 31);
            }
            a0[0] ^= (short)(it & 3);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)u0;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j] + acc2[j] + acc3[j];
    if (iters < 0) out[threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE> void run(int blocks_per_cu, int iters, float* out, unsigned long long* clk_d)
{
    const int grid = 256 * 16;
    size_t lds = 160 * 1024 / blocks_per_cu - 512;
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
    double iters_per_simd = (double)grid * 4 * iters / 1024.0;
    printf("%-48s waves/SIMD=%d  ms=%8.3f  clk=%7.1f MHz  cycles per loop body per SIMD = %8.2f\n", names[MODE], blocks_per_cu, ms, mhz, ms * 1e-3 * mhz * 1e6 / iters_per_simd);
}

int main()
{
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 1024)); CHECK(hipMalloc(&clk, 256 * 16 * 2 * sizeof(unsigned long long)));
    const int iters = 20000;
#define RUNALL(M) run<M>(1, iters, out, clk); run<M>(2, iters, out, clk); run<M>(4, iters, out, clk); run<M>(8, iters, out, clk);
    RUNALL(M_BF16_ONLY) RUNALL(M_TILE32_NOVALU) RUNALL(M_VALU_ONLY) RUNALL(M_TILE32) RUNALL(M_TILE16)
    return 0;
}
