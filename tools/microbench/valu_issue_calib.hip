// Calibration of the "vector issue utilisation" that bench.py reports (profiles/trace_kernel_valu_issue.json):
//   util = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * kernel cycles)
// Three kernels that do nothing but issue VALU instructions from 8 resident waves per SIMD -- v_fma_f32 (the fast class,
// 2.5 cycles per wave-instruction), v_alignbit_b32 (the 4.2-cycle class), and the two interleaved -- must read 1.0;
// whatever they read instead is the normalisation of the counter on this chip (tools/valu_issue.py divides by it).
// Run under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(1024) calib_fma(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0000001f, c = 1e-9f;
    for (int it = 0; it < iters; ++it)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 123.456f) out[0] = s;
}

__global__ void __launch_bounds__(1024) calib_alignbit(float* out, int iters)
{
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1;
    for (int it = 0; it < iters; ++it)
        asm volatile("v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %4, 31\n v_alignbit_b32 %2, %2, %4, 31\n v_alignbit_b32 %3, %3, %4, 31\n"
                     "v_alignbit_b32 %0, %0, %5, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %5, 31\n v_alignbit_b32 %3, %3, %5, 31\n"
                     : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));
    if (u0 + u1 + u2 + u3 == 123456u) out[0] = 1.0f;
}

__global__ void __launch_bounds__(1024) calib_mixed(float* out, int iters)
{
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001f, c = 1e-9f;
    for (int it = 0; it < iters; ++it)
        asm volatile("v_alignbit_b32 %0, %0, %4, 31\n v_fma_f32 %4, %4, %8, %9\n v_alignbit_b32 %1, %1, %5, 31\n v_fma_f32 %5, %5, %8, %9\n"
                     "v_alignbit_b32 %2, %2, %6, 31\n v_fma_f32 %6, %6, %8, %9\n v_alignbit_b32 %3, %3, %7, 31\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    if (u0 + u1 + u2 + u3 == 123456u) out[0] = a0 + a1 + a2 + a3;
}

int main()
{
    float* out;
    CHECK(hipMalloc(&out, 1024));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount * 2, iters = 400000;       // 2 x 1024 threads per CU = 8 waves per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
#define RUN(k) CHECK(hipEventRecord(e0)); k<<<grid, 1024>>>(out, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); \
    printf("%-16s %8.3f ms  %.3f cycles per wave-instruction per SIMD at 2.4 GHz\n", #k, ms, ms * 1e-3 * 2.4e9 / (8.0 * iters * 8.0));
    RUN(calib_fma) RUN(calib_alignbit) RUN(calib_mixed)
    CHECK(hipDeviceSynchronize());
    return 0;
}
