// ref_utils_wrap.cpp -- compiles the reference's OWN utils.cu (random, randomInUnitSphere, randomOnUnitDisk, delay,
// threadSetName; utils.cu:37-94) for the harness.  TEST INFRASTRUCTURE ONLY: linked into oracle/_ref/ref_harness, never
// into the product, never leaves the container that holds /root/reference.
//
// utils.cu does not compile on Linux as it stands: `reutrn` at :59, device-only symbols at :133,:147 and
// the CUDA error helpers at :154-163.  Everything this file adds is a stand-in for those -- none of it is on the path that
// the harness executes -- plus ONE hook: the reference draws with `_random( _gen )` (utils.cu:70), and that expression is
// redirected to refinject::draw() so that the harness can feed the loops of utils.cu:74-94 the shared stream.  The
// samplers keep their bodies but are renamed (pk::ref_randomInUnitSphere / pk::ref_randomOnUnitDisk); ref_utils_host.cpp
// wraps each in a scope that tells draw() which sampler is running.  This is a build with stand-ins for the CUDA toolkit
// the image lacks, so by the task's rule it pins nothing ("parity unpinned"); what it does is make the
// sampler KAT compare the restatement with the reference's own loops instead of with itself.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <random>
#include <thread>

#include "ref_inject.h"

// stand-ins for the device-only symbols of utils.cu:96-152 (dead code on the host)
namespace { struct Dim3Stub { unsigned x = 0, y = 0, z = 0; }; }
static Dim3Stub blockIdx, blockDim, threadIdx;
struct curandState { int unused; };
static inline void   curand_init(unsigned long long, int, int, curandState*) {}
static inline double curand_uniform_double(curandState*) { return 0.0; }
static inline const char* cudaGetErrorName(int) { return "cuda"; }
static inline const char* cudaGetErrorString(int) { return "not available"; }

#define reutrn return                                     /* utils.cu:59 */
// `static std::uniform_real_distribution<float> _random( 0.0f, 1.0f );` (two arguments: keeps a renamed, unused object)
// and `return _random( _gen );` (one argument: the hook)
#define REFINJ_SELECT(a, b, name, ...) name
#define REFINJ_DECL(a, b) _random_unused(a, b)
#define REFINJ_CALL(g) refinject::draw()
#define _random(...) REFINJ_SELECT(__VA_ARGS__, REFINJ_DECL, REFINJ_CALL, unused)(__VA_ARGS__)
#define randomInUnitSphere ref_randomInUnitSphere
#define randomOnUnitDisk ref_randomOnUnitDisk

#include REF_UTILS_CU
