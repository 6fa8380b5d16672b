// Microbenchmark: issue rate of SDWA forms on gfx950 (byte insert of the sign byte), and of a 32-value sign pack built from
// them (32 x v_and_b32_sdwa + 7 shifts + 4 v_bitop3) against 32 x v_alignbit_b32.  Same method as valu_rate2.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
enum { M_AND_SDWA, M_MOV_SDWA, M_PACK32_SDWA, M_PACK32_ALIGNBIT };
static const char* names[] = { "v_and_b32_sdwa (byte insert)", "v_mov_b32_sdwa (byte insert)", "pack 32 signs: sdwa + shifts + bitop3", "pack 32 signs: 32 x v_alignbit_b32" };
static const int per[] = { 8, 8, 1, 1 };

template <int MODE>
__global__ void __launch_bounds__(256) bench(unsigned* out, int iters, unsigned long long* clk)
{
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    unsigned w[32];
    for (int j = 0; j < 32; ++j) w[j] = (threadIdx.x * 2654435761u) ^ (j * 40503u);
    const unsigned m80 = 0x80u, M = 0x80808080u;
    unsigned acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_AND_SDWA) {
#define A1(D, K) "v_and_b32_sdwa " #D ", %8, %9 dst_sel:BYTE_" #K " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n"
            asm volatile(A1(%0, 0) A1(%1, 1) A1(%2, 2) A1(%3, 3) A1(%4, 0) A1(%5, 1) A1(%6, 2) A1(%7, 3)
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w[0]), "v"(m80));
        } else if (MODE == M_MOV_SDWA) {
#define V1(D, K) "v_mov_b32_sdwa " #D ", %8 dst_sel:BYTE_" #K " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n"
            asm volatile(V1(%0, 0) V1(%1, 1) V1(%2, 2) V1(%3, 3) V1(%4, 0) V1(%5, 1) V1(%6, 2) V1(%7, 3)
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(w[0]));
        } else if (MODE == M_PACK32_SDWA) {
            unsigned P[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                unsigned p = 0;
                asm volatile("v_and_b32_sdwa %0, %1, %5 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n"
                             "v_and_b32_sdwa %0, %2, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n"
                             "v_and_b32_sdwa %0, %3, %5 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n"
                             "v_and_b32_sdwa %0, %4, %5 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD\n"
                             : "+v"(p) : "v"(w[4 * g]), "v"(w[4 * g + 1]), "v"(w[4 * g + 2]), "v"(w[4 * g + 3]), "v"(m80));
                P[g] = p;
            }
            unsigned r = P[7];
#pragma unroll
            for (int g = 6; g >= 0; --g) r = (r >> 1) | P[g];       // lshr + or; pairs of these become one shift + v_bitop3 in the kernel
            acc ^= r;
            w[0] += acc; w[5] ^= acc;
        } else {
            unsigned r = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) r = __builtin_amdgcn_alignbit(r, w[j], 31);
            acc ^= r;
            w[0] += acc; w[5] ^= acc;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned s = u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 + acc + M;
    if (iters < 0) out[threadIdx.x] = s;
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
    double per_simd = (double)grid * 4 * iters * per[MODE] / 1024.0;
    printf("%-44s waves/SIMD=%d  ms=%8.3f  clk=%7.1f MHz  cycles per %s per SIMD = %8.3f\n", names[MODE], blocks_per_cu, ms, mhz, per[MODE] == 1 ? "32-value pack" : "inst", ms * 1e-3 * mhz * 1e6 / per_simd);
}

int main()
{
    unsigned* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 1024)); CHECK(hipMalloc(&clk, 256 * 16 * 2 * sizeof(unsigned long long)));
    const int iters = 20000;
#define RUN(M) run<M>(4, iters, out, clk); run<M>(8, iters, out, clk);
    RUN(M_AND_SDWA) RUN(M_MOV_SDWA) RUN(M_PACK32_SDWA) RUN(M_PACK32_ALIGNBIT)
    return 0;
}
