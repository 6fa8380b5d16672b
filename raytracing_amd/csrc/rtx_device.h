// rtx_device.h -- shared declarations of the trace kernels (device side) and the C-ABI host code.
#pragma once

#include "../../include/rtx.h"
#include "rtx_plan.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtx
{

// Tuning constants of the candidate filter ("conservative filter + exact re-test").
// The filter works on the inflated quantities r^2 (1 + kEps) and h^2 (1 + kEps)^2; kEps = 2^-17 = 128 ulp covers
// the <= 34 ulp that the reference's own rounding of b*b - a*c can move the sign of the discriminant.
constexpr float    kEps        = 7.62939453125e-06f;  // 2^-17
constexpr uint32_t kMaxBlock   = 512;                  // threads per workgroup (launch bound); the block size is chosen per launch
constexpr uint32_t kWaveBatch  = 64;                    // smallest reservation of work items (TraceParams::wave_batch)
constexpr uint32_t kWaveLdsBytes = 4608;               // per wave: 64 ray records (2 KB), 64 result slots (512 B), 256 candidate entries (1 KB), 128 sign-word entries (1 KB)
constexpr uint32_t kNoHit      = 0xFFFFFFFFu;
constexpr uint32_t kWaveLdsBytesGrid = 2304 + 13 * 256;    // grid variant: phase (D)'s cooperative sampling (2304 B), then 13 parked words per lane
constexpr uint32_t kChunkTabBytes = 256;               // the LDS copy of the boundaries, in front of everything else
// MFMA filter: absolute margin gamma * (|C|^2 + r^2 + |K'|) + kMfmaAbsMargin / scale^2 per sphere and gamma * |o|^2 per
// ray, gamma = 2^-16 = 256 ulp (ulp = 2^-24).  Budget, in ulp of (|C|^2 + |o|^2), for the single quadratic form: a monomial
// is x1 p1 + x1 p2 + x2 p1 with both factors split to 2^-22, i.e. 12 ulp of |monomial x coefficient|; the coefficients
// themselves (d'_i d'_j, o.d' d'_i - o_i, |o|^2 - (o.d')^2) carry <= 5 ulp from d' and o.d'.  Quadratic part: 17 ulp of
// (sum |C_i d'_i|)^2 <= 17 |C|^2; linear part: 17 ulp of 4 |C||o| <= 34; K'' 4; constant 8 |o|^2; accumulation of the two
// chained v_mfma_f32_32x32x16_f16 (<= 5.7 ulp of sum |terms| each, tools/microbench/mfma_f16_error.hip, sum |terms| <=
// 4 (|C|^2 + |o|^2)) 46; the reference's own rounding of b*b - a*c moves the sign within 34.  Sum 143 of 256.  Worst case of
// a model instead of the measurement -- exact f16 x f16 products, 16 additions per MFMA each rounded to nearest in any order:
// 16 ulp per MFMA, 128 for the two -- gives 225 of 256: the margin also holds there ("filter margins").
// kMfmaAbsMargin covers f16 subnormals: second parts below 2^-14 are rounded to multiples of 2^-24, which against a
// partner of up to 2^16 is 2^-9 (in scale^2 units) per slot pair; ten monomials.
constexpr double   kMfmaAbsMargin = 0.03125;                         // 2^-5
constexpr float    kGammaRay   = 1.52587890625e-05f * 1.0625f;   // 2^-16 plus room for the rounding of |o|^2 itself
constexpr double   kGammaSphere = 1.52587890625e-05;
enum FilterKind { kFilterValu = 0, kFilterMfma = 1, kFilterGrid = 2, kFilterAuto = 3 };   // kFilterAuto is a host-side choice, never a kernel variant
constexpr uint32_t kAutoGridSpheres = 257u;
// where the filter's sphere table lives: read through L2 by every wave, resident in LDS, or (MFMA filter) streamed
// through a double-buffered LDS stage that the waves of a workgroup fill together, 128 spheres at a time
enum TableMode { kTableGlobal = 0, kTableLds = 1, kTableStream = 2 };
#ifndef RTX_STREAM_GROUPS
#define RTX_STREAM_GROUPS 2
#endif
constexpr uint32_t kStreamGroups = RTX_STREAM_GROUPS;             // 128-sphere groups per staged chunk: one workgroup barrier per chunk
constexpr uint32_t kStreamStageBytes = 2 * kStreamGroups * 8192;  // two buffers of one chunk (4 tiles x 2 KB per group)
constexpr uint32_t kMaxLdsSpheres = 2048;              // VALU filter: 16 B/sphere table stays LDS-resident up to here
#ifndef RTX_MAX_LDS_MFMA
#define RTX_MAX_LDS_MFMA 1024
#endif
constexpr uint32_t kMaxLdsSpheresMfma = RTX_MAX_LDS_MFMA;          // MFMA filter: 64 B/sphere (+ 16 B exact record)

__host__ __device__ constexpr uint32_t wave_lds_bytes(int filter) { return filter == kFilterGrid ? kWaveLdsBytesGrid : kWaveLdsBytes; }

enum StatSlot { kStSamples = 0, kStRays, kStMiss, kStAbsorb, kStDepth, kStCandidates, kStFilterMiss, kStCount };

struct TraceParams {
    const float4* hot;      // n_pad x (cx, cy, cz, -(r^2 (1+eps)))  -- the VALU filter's view of a sphere
    const float4* hot_mfma; // (n_pad_mfma/32 + 1) tiles x 2 MFMAs x 64 lanes x 16 B: f16 A operands (ten monomials per sphere) of the MFMA filter
    uint32_t      n_pad_mfma;
    float         mfma_scale; // power of two applied to the sphere side of the MFMA filter (f16 range)
    GridDesc      grid;       // RTX_FILTER_GRID: uniform grid over the table (rtx_plan.h); a cell's entries are stored tier by tier; the walk reads one 32-byte record per trip
    const float4*   grid_cells;  // 2 per cell: (first entry, index of its sphere, entries of tier 0 | up to tier 1 << 16, entries up to tier 2), (that sphere's cx, cy, cz, r)
    const float4*   grid_recs;   // 2 per entry: (cx, cy, cz, r) of the registered sphere -- a copy, so that a test is one fetch --, (its index, -, -, -)
    const uint32_t* grid_big;  // grid.n_big spheres tested for every ray
    const float4* exact;    // n x (cx, cy, cz, r)                   -- sphere.h:24-28 geometry, as uploaded
    const uint4*  mat;      // 2n: [2i] = (type, albedo r, g, b), [2i+1] = (blur, ior, 1.0 / (double)radius: low word, high word), float bits;
                            // glass (material.cu:69-105 reads neither albedo nor blur): albedo r holds ((1 - ior) / (1 + ior))^2, blur 1.0f / ior
    uint32_t      n, n_pad;
    rtx_camera    cam;
    uint32_t      cols, rows, npix;
    double        inv_cols, inv_rows;   // 1.0 / (double)cols, 1.0 / (double)rows (phase (B)'s two divisions)
    uint32_t      spp_begin, spp_count, chunk_spp, nchunks;   // chunk_spp != 0: uniform chunks; 0: chunk k covers samples [chunk_tab[k], chunk_tab[k+1]) of the range
    uint32_t      chunk_tab[kMaxPlanChunks + 1];
    uint32_t      max_depth, key0, key1, flags;
    uint32_t      n_items;
    uint32_t      wave_batch; // work items a wave reserves per trip to the global counter (host: about 512 samples' worth)
    uint32_t      tail_guard; // the last tail_guard items of the list are reserved on demand instead of ahead of time
    uint32_t      walk_park;  // grid variant: park the walks of the last walk_park rays of a wave (0: every walk runs to its end in its own iteration)
    const uint32_t* order;    // NULL: item it = (chunk it / npix, pixel it % npix); else cost-ordered batches of 64 items: items 64 b .. 64 b + 63 are
                              // chunk order[b] >> 22 of the pixels 64 s .. 64 s + 63 of strip s = order[b] & 0x3FFFFF
    float4*       partials;   // n_items x (sum r, g, b, rays as uint bits): item = chunk * npix + pixel
    float4*       att_stack;  // RTX_FLAG_RECURSIVE only: max_depth x (grid * block) attenuations
    uint32_t*     primary;    // NULL or spp_count * npix
    uint32_t*     work_counter;
    unsigned long long* stats; // kStCount counters
#ifdef RTX_WAVE_TIMES
    unsigned long long* wave_times;   // diagnostic build: 5 words per wave
#endif
#ifdef RTX_WALK_STATS
    unsigned long long* walk_stats;   // diagnostic build (grid walk): 64 counters, see trace_kernel.hip
#endif
};

hipError_t launch_trace(const TraceParams& p, int filter, int table_mode, bool validate, uint32_t grid, uint32_t block, size_t lds_bytes, hipStream_t stream);
int  trace_occupancy(int filter, int table_mode, size_t lds_bytes, uint32_t block);
size_t trace_lds_bytes(int filter, int table_mode, uint32_t n, uint32_t n_pad, uint32_t n_pad_mfma, uint32_t block);
void launch_reduce(const float4* partials, float* accum, uint32_t* cost, unsigned long long* cost_total, uint32_t npix, uint32_t nchunks, bool accumulate, hipStream_t stream);
void launch_quotient_check(uint32_t seed, uint32_t mode, uint32_t blocks, uint32_t rounds, unsigned long long* bad, hipStream_t stream);
void launch_fill(float* accum, uint32_t n_floats, float value, bool accumulate, hipStream_t stream);
void launch_resolve(const float* accum, uint32_t* fb, uint32_t cols, uint32_t rows, uint32_t spp_total, uint32_t debug_tile, hipStream_t stream);

} // namespace rtx
