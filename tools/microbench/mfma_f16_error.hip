// How accurately does v_mfma_f32_32x32x16_f16 sum its 16 exact products + C (and is the operand layout the one assumed)?  Reports max |D - exact| in units of
// ulp(2^-24) * (sum |a_k b_k| + |c|), over random and cancellation-heavy inputs (exact reference in double).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// A: [tile][32 rows][16 k] f16, B: [tile][16 k][32 cols] f16, C/D: [tile][32 rows][32 cols] f32
// lane l: A row l%32, k 8(l/32)..+7; B col l%32, same k; D element v: row 8(v/4) + 4(l/32) + v%4, col l%32
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D, int tiles)
{
    int t = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (t >= tiles) return;
    int l = threadIdx.x & 63, r = l & 31, hf = l >> 5;
    h8v a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[((size_t)t * 32 + r) * 16 + 8 * hf + j]; b[j] = B[((size_t)t * 16 + 8 * hf + j) * 32 + r]; }
    f16v c;
    for (int v = 0; v < 16; ++v) c[v] = C[((size_t)t * 32 + 8 * (v / 4) + 4 * hf + v % 4) * 32 + r];
    f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D[((size_t)t * 32 + 8 * (v / 4) + 4 * hf + v % 4) * 32 + r] = d[v];
}


int main()
{
    const int tiles = 4096;
    std::vector<_Float16> A((size_t)tiles * 32 * 16), B((size_t)tiles * 16 * 32);
    std::vector<float> C((size_t)tiles * 1024), D((size_t)tiles * 1024);
    std::mt19937 gen(1);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    for (int mode = 0; mode < 4; ++mode) {
        for (int t = 0; t < tiles; ++t) {
            for (int r = 0; r < 32; ++r) for (int kk = 0; kk < 16; ++kk) {
                float v = U(gen);
                if (mode == 1) v *= std::ldexp(1.0f, -(kk % 2) * 11);           // split-like magnitudes 1, 2^-11
                if (mode == 2) v *= std::ldexp(1.0f, (int)(gen() % 20) - 10);    // wide dynamic range
                if (mode == 3) v = (kk & 1 ? -1.0f : 1.0f) * (1.0f + 0.01f * U(gen)) * 100.0f; // heavy cancellation
                A[((size_t)t * 32 + r) * 16 + kk] = (_Float16)v;
            }
            for (int kk = 0; kk < 16; ++kk) for (int c = 0; c < 32; ++c) {
                float v = U(gen);
                if (mode == 1) v *= std::ldexp(1.0f, -((kk + 1) % 2) * 11);
                if (mode == 2) v *= std::ldexp(1.0f, (int)(gen() % 20) - 10);
                if (mode == 3) v = (1.0f + 0.01f * U(gen));
                B[((size_t)t * 16 + kk) * 32 + c] = (_Float16)v;
            }
            for (int i = 0; i < 1024; ++i) C[(size_t)t * 1024 + i] = (mode == 3 ? 50.0f : 1.0f) * U(gen);
        }
        _Float16 *dA, *dB; float *dC, *dD;
        CHECK(hipMalloc(&dA, A.size() * 2)); CHECK(hipMalloc(&dB, B.size() * 2)); CHECK(hipMalloc(&dC, C.size() * 4)); CHECK(hipMalloc(&dD, D.size() * 4));
        CHECK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
        k<<<tiles / 4, 256>>>(dA, dB, dC, dD, tiles);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, worst_rel_result = 0; size_t exact_rne = 0, total = 0;
        for (int t = 0; t < tiles; ++t) for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            double s = C[(size_t)t * 1024 + r * 32 + c], sabs = std::fabs(s);
            for (int kk = 0; kk < 16; ++kk) {
                double p = (double)(float)A[((size_t)t * 32 + r) * 16 + kk] * (double)(float)B[((size_t)t * 16 + kk) * 32 + c];
                s += p; sabs += std::fabs(p);
            }
            double got = D[(size_t)t * 1024 + r * 32 + c];
            double err = std::fabs(got - s) / (std::ldexp(1.0, -24) * sabs);
            if (err > worst) worst = err;
            if ((float)s == (float)got) exact_rne++;
            total++;
            double rel = std::fabs(got - s) / (std::ldexp(1.0, -24) * std::fabs(s) + 1e-300);
            if (rel > worst_rel_result) worst_rel_result = rel;
        }
        printf("mode %d: max err = %.4f x 2^-24 x sum|terms| ; max err = %.2f x 2^-24 x |result| ; correctly rounded results: %.4f %%\n", mode, worst, worst_rel_result, 100.0 * exact_rne / total);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    }
    return 0;
}
