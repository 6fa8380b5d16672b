// Microbenchmark: VALU issue rates on gfx950 for the instructions the trace
// kernel's inner loop is built from. Prints instructions/clk/SIMD assuming the
// measured in-kernel clock. Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum { M_FMA, M_PKFMA, M_PKMUL, M_PKADD, M_MAD64, M_MULLO, M_MULHI, M_ALIGNBIT, M_SQRT, M_RCP, M_FMA_SGPR, M_FMA_LDS, M_CMP, M_MIX11, M_PKMIX, M_COUNT };
static const char* names[] = { "v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_alignbit_b32", "v_sqrt_f32", "v_rcp_f32", "v_fma_f32(sgpr src)", "11 v_fma + 1 ds_read_b128 bcast", "v_cmp_lt_f32", "3sub+7fma+alignbit (sphere test)", "pk sphere test x2 (6 pk + 2 alignbit)" };
// instructions per loop body for each mode
#define UNR 8

template <int MODE>
__global__ void __launch_bounds__(1024) bench(float* out, int iters, unsigned long long* clk)
{
    __shared__ f4 lds[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) lds[i] = f4{ (float)i, 1.0f, 2.0f, 3.0f };
    __syncthreads();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0000001f, c = 1e-9f;
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, p4 = { a1, a0 }, p5 = { a3, a2 }, p6 = { a5, a4 }, p7 = { a7, a6 };
    f2 pb = { b, b }, pc = { c, c };
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    unsigned long long w0 = threadIdx.x, w1 = w0 + 7, w2 = w0 + 9, w3 = w0 + 11;
    float sb; // scalar
    asm volatile("s_mov_b32 %0, 0x3f800001" : "=s"(sb));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_FMA) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == M_FMA_SGPR) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sb), "v"(c));
        } else if (MODE == M_PKFMA) {
            asm volatile(
                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
        } else if (MODE == M_PKMUL) {
            asm volatile(
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));
        } else if (MODE == M_PKADD) {
            asm volatile(
                "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
        } else if (MODE == M_MAD64) {
            asm volatile(
                "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(u0), "v"(u1) : "vcc");
        } else if (MODE == M_MULLO) {
            asm volatile(
                "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x9E3779B9u));
        } else if (MODE == M_MULHI) {
            asm volatile(
                "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x9E3779B9u));
        } else if (MODE == M_ALIGNBIT) {
            asm volatile(
                "v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %4, 31\n v_alignbit_b32 %2, %2, %4, 31\n v_alignbit_b32 %3, %3, %4, 31\n"
                "v_alignbit_b32 %0, %0, %5, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %5, 31\n v_alignbit_b32 %3, %3, %5, 31\n"
                : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));
        } else if (MODE == M_SQRT) {
            asm volatile(
                "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == M_RCP) {
            asm volatile(
                "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == M_CMP) {
            asm volatile(
                "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n"
                "v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0\n"
                :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
        } else if (MODE == M_FMA_LDS) {
            // one broadcast 16-byte LDS read per 11 VALU (the real loop's ratio), 2 "spheres" per body
            f4 s0 = lds[(it * 2) & 511], s1 = lds[(it * 2 + 1) & 511];
            asm volatile(
                "v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %9, %1\n v_sub_f32 %2, %10, %2\n v_fma_f32 %3, %0, %0, %11\n v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n"
                "v_mul_f32 %4, %0, %5\n v_fma_f32 %4, %1, %6, %4\n v_fma_f32 %4, %2, %7, %4\n v_fma_f32 %3, -%4, %4, %3\n v_alignbit_b32 %5, %5, %3, 31\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0.x), "v"(s0.y), "v"(s0.z), "v"(s0.w));
            asm volatile(
                "v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %9, %1\n v_sub_f32 %2, %10, %2\n v_fma_f32 %3, %0, %0, %11\n v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n"
                "v_mul_f32 %4, %0, %5\n v_fma_f32 %4, %1, %6, %4\n v_fma_f32 %4, %2, %7, %4\n v_fma_f32 %3, -%4, %4, %3\n v_alignbit_b32 %5, %5, %3, 31\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s1.x), "v"(s1.y), "v"(s1.z), "v"(s1.w));
        } else if (MODE == M_MIX11) {
            // the sphere-test instruction mix with all operands in VGPRs, two independent tests interleaved
            asm volatile(
                "v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %9, %1\n v_sub_f32 %2, %8, %2\n v_fma_f32 %3, %0, %0, %9\n v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n"
                "v_mul_f32 %4, %0, %5\n v_fma_f32 %4, %1, %6, %4\n v_fma_f32 %4, %2, %7, %4\n v_fma_f32 %3, -%4, %4, %3\n v_alignbit_b32 %5, %5, %3, 31\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (MODE == M_PKMIX) {
            // packed: two spheres per lane per instruction: 3 pk_add(sub) + 3 pk_fma (c) + 1 pk_mul + 2 pk_fma (h) + 1 pk_fma (disc) + 2 alignbit = 12 inst / 2 spheres
            asm volatile(
                "v_pk_add_f32 %0, %4, %0 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %5, %1 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %4, %2 neg_lo:[0,1] neg_hi:[0,1]\n"
                "v_pk_fma_f32 %3, %0, %0, %5\n v_pk_fma_f32 %3, %1, %1, %3\n v_pk_fma_f32 %3, %2, %2, %3\n"
                "v_pk_mul_f32 %6, %0, %4\n v_pk_fma_f32 %6, %1, %5, %6\n v_pk_fma_f32 %6, %2, %4, %6\n v_pk_fma_f32 %3, %6, %6, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc), "v"(p4));
            asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %2, 31\n" : "+v"(u0) : "v"(p3.x), "v"(p3.y));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x + (float)(u0 + u1 + u2 + u3) + (float)(w0 + w1 + w2 + w3);
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

static int insts_per_body(int m) { if (m == M_FMA_LDS) return 22; if (m == M_MIX11) return 11; if (m == M_PKMIX) return 12; return 8; }

template <int MODE> void run(int threads, int blocks_per_cu, int iters, float* out, unsigned long long* clk_d)
{
    int ncu = 256; int grid = ncu * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    bench<MODE><<<grid, threads>>>(out, iters / 10, clk_d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    bench<MODE><<<grid, threads>>>(out, iters, clk_d);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> clk(grid * 2);
    CHECK(hipMemcpy(clk.data(), clk_d, grid * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; for (int i = 0; i < grid; ++i) { cyc += clk[2 * i]; rt += clk[2 * i + 1]; }
    cyc /= grid; rt /= grid;
    double mhz = cyc / rt * 100.0;
    int waves_per_simd = threads / 64 * blocks_per_cu / 4;
    double inst_per_wave = (double)iters * insts_per_body(MODE);
    // cycles per instruction per SIMD = cycles / (instructions issued on that SIMD)
    double cpi = cyc / (inst_per_wave * (waves_per_simd > 0 ? waves_per_simd : 1));
    printf("%-40s thr=%4d blk/CU=%d waves/SIMD=%d  ms=%8.3f  clk=%7.1f MHz  cyc/inst/SIMD=%6.3f\n", names[MODE], threads, blocks_per_cu, waves_per_simd, ms, mhz, cpi);
}

int main()
{
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 1024)); CHECK(hipMalloc(&clk, 256 * 8 * 2 * sizeof(unsigned long long)));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int iters = 200000;
#define RUNALL(M) run<M>(256, 1, iters, out, clk); run<M>(512, 1, iters, out, clk); run<M>(1024, 1, iters, out, clk); run<M>(1024, 2, iters, out, clk);
    RUNALL(M_FMA) RUNALL(M_PKFMA) RUNALL(M_PKMUL) RUNALL(M_PKADD) RUNALL(M_FMA_SGPR) RUNALL(M_MAD64) RUNALL(M_MULLO) RUNALL(M_MULHI)
    RUNALL(M_ALIGNBIT) RUNALL(M_CMP) RUNALL(M_SQRT) RUNALL(M_RCP) RUNALL(M_MIX11) RUNALL(M_FMA_LDS) RUNALL(M_PKMIX)
    return 0;
}
