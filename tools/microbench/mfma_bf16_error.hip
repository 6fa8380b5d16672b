// How accurately does v_mfma_f32_16x16x32_bf16 sum its 32 exact products + C?  Reports max |D - exact| in units of
// ulp(2^-24) * (sum |a_k b_k| + |c|), over random and cancellation-heavy inputs (exact reference in double).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

// A: [tile][16 rows][32 k] bf16, B: [tile][32 k][16 cols] bf16, C/D: [tile][16 rows][16 cols] f32
__global__ void k(const unsigned short* A, const unsigned short* B, const float* C, float* D, int tiles)
{
    int t = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (t >= tiles) return;
    int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    s8v a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[((size_t)t * 16 + r) * 32 + 8 * g + j]; b[j] = (short)B[((size_t)t * 32 + 8 * g + j) * 16 + r]; }
    f4v c;
    for (int j = 0; j < 4; ++j) c[j] = C[((size_t)t * 16 + 4 * g + j) * 16 + r];
    f4v d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[((size_t)t * 16 + 4 * g + j) * 16 + r] = d[j];
}

static unsigned short f2bf_trunc(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main()
{
    const int tiles = 8192;
    std::vector<unsigned short> A((size_t)tiles * 16 * 32), B((size_t)tiles * 32 * 16);
    std::vector<float> C((size_t)tiles * 256), D((size_t)tiles * 256);
    std::mt19937 gen(1);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    for (int mode = 0; mode < 4; ++mode) {
        for (int t = 0; t < tiles; ++t) {
            for (int r = 0; r < 16; ++r) for (int kk = 0; kk < 32; ++kk) {
                float v = U(gen);
                if (mode == 1) v *= std::ldexp(1.0f, -(kk % 3) * 8);            // split-like magnitudes 1, 2^-8, 2^-16
                if (mode == 2) v *= std::ldexp(1.0f, (int)(gen() % 24) - 12);    // wide dynamic range
                if (mode == 3) v = (kk & 1 ? -1.0f : 1.0f) * (1.0f + 0.01f * U(gen)) * 100.0f; // heavy cancellation
                A[((size_t)t * 16 + r) * 32 + kk] = f2bf_trunc(v);
            }
            for (int kk = 0; kk < 32; ++kk) for (int c = 0; c < 16; ++c) {
                float v = U(gen);
                if (mode == 1) v *= std::ldexp(1.0f, -((kk + 1) % 3) * 8);
                if (mode == 2) v *= std::ldexp(1.0f, (int)(gen() % 24) - 12);
                if (mode == 3) v = (1.0f + 0.01f * U(gen));
                B[((size_t)t * 32 + kk) * 16 + c] = f2bf_trunc(v);
            }
            for (int i = 0; i < 256; ++i) C[(size_t)t * 256 + i] = (mode == 3 ? 50.0f : 1.0f) * U(gen);
        }
        unsigned short *dA, *dB; float *dC, *dD;
        CHECK(hipMalloc(&dA, A.size() * 2)); CHECK(hipMalloc(&dB, B.size() * 2)); CHECK(hipMalloc(&dC, C.size() * 4)); CHECK(hipMalloc(&dD, D.size() * 4));
        CHECK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
        k<<<tiles / 4, 256>>>(dA, dB, dC, dD, tiles);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, worst_rel_result = 0; size_t exact_rne = 0, total = 0;
        for (int t = 0; t < tiles; ++t) for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
            double s = C[(size_t)t * 256 + r * 16 + c], sabs = std::fabs(s);
            for (int kk = 0; kk < 32; ++kk) {
                double p = (double)bf2f(A[((size_t)t * 16 + r) * 32 + kk]) * (double)bf2f(B[((size_t)t * 32 + kk) * 16 + c]);
                s += p; sabs += std::fabs(p);
            }
            double got = D[(size_t)t * 256 + r * 16 + c];
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
