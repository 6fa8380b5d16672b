// Microbenchmark v2: aggregate VALU throughput (wall clock, saturated chip) vs waves per SIMD.
// grid = 256 CUs x 16 blocks of 256 threads; occupancy is limited by the dynamic LDS size.
// cyc/inst/SIMD = wall * clock / (instructions issued per SIMD); clock from s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
enum { M_FMA, M_PKFMA, M_MAD64, M_MULLO, M_ALIGNBIT, M_SQRT, M_CNDMASK, M_MIX11, M_MIX11_LDS, M_XOR, M_MFMA32, M_MFMA16, M_MFMA_VALU, M_COUNT };
static const char* names[] = { "v_fma_f32", "v_pk_fma_f32", "v_mad_u64_u32", "v_mul_lo_u32", "v_alignbit_b32", "v_sqrt_f32", "v_cndmask_b32", "sphere mix (11 valu)", "sphere mix + ds_read_b128 bcast", "v_xor_b32", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x4_f32", "4 mfma32 + 32 valu (filter tile)" };
static int ipb[] = { 8, 8, 8, 8, 8, 8, 8, 11, 22, 8, 4, 4, 36 };
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) bench(float* out, int iters, unsigned long long* clk)
{
    extern __shared__ f4 lds[];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) lds[i] = f4{ (float)i, 1.0f, 2.0f, 3.0f };
    __syncthreads();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0000001f, c = 1e-9f;
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, p4 = { a1, a0 }, p5 = { a3, a2 }, p6 = { a5, a4 }, p7 = { a7, a6 };
    f2 pb = { b, b }, pc = { c, c };
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    unsigned long long w0 = threadIdx.x, w1 = w0 + 7, w2 = w0 + 9, w3 = w0 + 11;
    f16v acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v q0 = {0}, q1 = {0}, q2 = {0}, q3 = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_FMA) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == M_PKFMA) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
        } else if (MODE == M_MAD64) {
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                         "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                         : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(u0), "v"(u1) : "vcc");
        } else if (MODE == M_MULLO) {
            asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                         "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x9E3779B9u));
        } else if (MODE == M_ALIGNBIT) {
            asm volatile("v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %4, 31\n v_alignbit_b32 %2, %2, %4, 31\n v_alignbit_b32 %3, %3, %4, 31\n"
                         "v_alignbit_b32 %0, %0, %5, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %5, 31\n v_alignbit_b32 %3, %3, %5, 31\n"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));
        } else if (MODE == M_XOR) {
            asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n"
                         "v_xor_b32 %0, %0, %5\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %5\n v_xor_b32 %3, %3, %5\n"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));
        } else if (MODE == M_SQRT) {
            asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                         "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == M_CNDMASK) {
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        } else if (MODE == M_MIX11) {
            asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %9, %1\n v_sub_f32 %2, %8, %2\n v_fma_f32 %3, %0, %0, %9\n v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n"
                         "v_mul_f32 %4, %0, %5\n v_fma_f32 %4, %1, %6, %4\n v_fma_f32 %4, %2, %7, %4\n v_fma_f32 %3, -%4, %4, %3\n v_alignbit_b32 %5, %5, %3, 31\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == M_MIX11_LDS) {
            f4 s0 = lds[(it * 2) & 511], s1 = lds[(it * 2 + 1) & 511];
            asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %9, %1\n v_sub_f32 %2, %10, %2\n v_fma_f32 %3, %0, %0, %11\n v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n"
                         "v_mul_f32 %4, %0, %5\n v_fma_f32 %4, %1, %6, %4\n v_fma_f32 %4, %2, %7, %4\n v_fma_f32 %3, -%4, %4, %3\n v_alignbit_b32 %5, %5, %3, 31\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0.x), "v"(s0.y), "v"(s0.z), "v"(s0.w));
            asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %9, %1\n v_sub_f32 %2, %10, %2\n v_fma_f32 %3, %0, %0, %11\n v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n"
                         "v_mul_f32 %4, %0, %5\n v_fma_f32 %4, %1, %6, %4\n v_fma_f32 %4, %2, %7, %4\n v_fma_f32 %3, -%4, %4, %3\n v_alignbit_b32 %5, %5, %3, 31\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s1.x), "v"(s1.y), "v"(s1.z), "v"(s1.w));
        } else if (MODE == M_MFMA32) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, a3, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4, a5, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a6, a7, acc3, 0, 0, 0);
        } else if (MODE == M_MFMA16) {
            q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, q0, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, a3, q1, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4, a5, q2, 0, 0, 0);
            q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a6, a7, q3, 0, 0, 0);
        } else if (MODE == M_MFMA_VALU) {
            // one 32 rays x 32 spheres filter tile: H = 2 mfma, W = fma(-h,h,oo) (16 valu), C = 2 mfma on top of W, 16 alignbit
            f16v h = {0};
            h = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, h, 0, 0, 0);
            h = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, a3, h, 0, 0, 0);
            f16v w;
#pragma unroll
            for (int j = 0; j < 16; ++j) w[j] = __builtin_fmaf(-h[j], h[j], a4);
            w = __builtin_amdgcn_mfma_f32_32x32x2f32(a5, a6, w, 0, 0, 0);
            w = __builtin_amdgcn_mfma_f32_32x32x2f32(a7, a0, w, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) u0 = __builtin_amdgcn_alignbit(u0, __float_as_uint(w[j]), 31);
            a0 += 1e-9f;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x + (float)(u0 + u1 + u2 + u3) + (float)(w0 + w1 + w2 + w3);
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j] + acc2[j] + acc3[j];
    s += q0[0] + q1[1] + q2[2] + q3[3];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE> void run(int blocks_per_cu, int iters, float* out, unsigned long long* clk_d)
{
    const int grid = 256 * 16;
    size_t lds = 160 * 1024 / blocks_per_cu - 512;   // limits residency to blocks_per_cu
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
    double inst_per_simd = (double)grid * 4 /*waves*/ * iters * ipb[MODE] / 1024.0;
    double cpi = ms * 1e-3 * mhz * 1e6 / inst_per_simd;
    printf("%-36s waves/SIMD=%d  ms=%8.3f  clk=%7.1f MHz  cyc/inst/SIMD=%7.3f\n", names[MODE], blocks_per_cu, ms, mhz, cpi);
}

int main()
{
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 1024)); CHECK(hipMalloc(&clk, 256 * 16 * 2 * sizeof(unsigned long long)));
    const int iters = 20000;
#define RUNALL(M) run<M>(1, iters, out, clk); run<M>(2, iters, out, clk); run<M>(4, iters, out, clk); run<M>(6, iters, out, clk); run<M>(8, iters, out, clk);
#ifdef ONLY_MFMA
    RUNALL(M_MFMA_VALU) return 0;
#endif
    RUNALL(M_FMA) RUNALL(M_PKFMA) RUNALL(M_MAD64) RUNALL(M_MULLO) RUNALL(M_ALIGNBIT) RUNALL(M_XOR) RUNALL(M_SQRT) RUNALL(M_CNDMASK) RUNALL(M_MIX11) RUNALL(M_MIX11_LDS)
    RUNALL(M_MFMA32) RUNALL(M_MFMA16) RUNALL(M_MFMA_VALU)
    return 0;
}
