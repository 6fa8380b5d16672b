// trace_kernel.hip -- the per-pixel trace loop of avneic/RayTracing as hand-written HIP for gfx950 (CDNA4).
//
// What it replaces (reference file:line):
//   _renderJob's sample loop           raytracer.cpp:157-186      -> persistent waves, one path per lane, lane refill
//   thread_pool tile jobs              raytracer.cpp:93-127       -> work items (sample chunk, pixel) drawn from one
//                                                                   global counter, 64 at a time per wave
//   Camera::getRay                     camera.h:75-86             -> phase (B) of the loop
//   _color / _color_recursive          raytracer.cpp:213-273      -> the bounce state kept in each lane
//   _sceneHit + sphereHit              raytracer.cpp:285-304,
//                                      sphere.cpp:68-102          -> phase (C): conservative candidate filter over the
//                                                                   whole sphere table + exact re-test of the few
//                                                                   candidates in wave-wide batches
//   materialScatter and helpers        material.cu:17-138         -> phase (D)
//   random / samplers                  utils.cu:65-94             -> Philox4x32-10 counter stream, rejection
//                                                                   sampling shared by all lanes of the wave
//
// Two candidate filters share everything else (select with rtx_set_filter / RTX_FILTER):
//   kFilterValu  : per lane, per sphere: oc = C - o, h = oc.d', q = |oc|^2 - r^2(1+eps), sign(q - h^2) -> 11 VALU ops
//   kFilterMfma  : the same test expanded into one quadratic form of the sphere -- ten monomials (C_i C_j, C_i, 1) against
//                  ten per-ray coefficients -- and evaluated for 32 spheres x 32 rays by two chained
//                  v_mfma_f32_32x32x16_f16; every factor is split into two f16 parts (22 bits) behind a per-scene
//                  power-of-two scale, and an absolute margin proportional to |C|^2 + |o|^2 keeps it conservative
//                  ("filter margins").  The VALU only captures the signs.
//
// Numerics: everything that decides a result (camera ray, the exact sphere test, hit point, normal, scatter,
// background, accumulation) is evaluated in the reference's operation order with contraction OFF, IEEE division and
// square root, so a path is bit-identical to the CPU restatement of the reference.  Only the candidate filter uses
// FMAs, MFMAs and a reciprocal square root; it is conservative (never drops a sphere the exact test would accept),
// so it cannot change a result -- only how many exact tests run.
#include <cstddef>
#include "rtx_device.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace rtx
{

// ---------------------------------------------------------------------------------------------------------------
// small vector helpers, each the image of one operator of vector_cuda.h (association order matters)
struct V3 { float x, y, z; };
__device__ __forceinline__ V3    mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3    ld3(const float* p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3    add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }        // :99-102
__device__ __forceinline__ V3    sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }        // :104-107
__device__ __forceinline__ V3    mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }        // :109-112
__device__ __forceinline__ V3    scale(float t, V3 v) { return mk(t * v.x, t * v.y, t * v.z); }         // :119-122, :129-132
__device__ __forceinline__ V3    divs(V3 v, float t) { return mk(v.x / t, v.y / t, v.z / t); }          // :124-127
__device__ __forceinline__ V3    neg(V3 v) { return mk(-v.x, -v.y, -v.z); }                             // :30
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }    // :68-71
__device__ __forceinline__ float sqlen(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }              // :42

// ---------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Random123 constants).  key = seed, counter = (slot, sample, pixel, 0).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // hi ^ c ^ key in one instruction (v_bitop3_b32, truth table 0x96 = a ^ b ^ c); the compiler emits two v_xor
        uint32_t n0, n2;
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n0) : "v"((uint32_t)(p1 >> 32)), "v"(c1), "s"(k0));
        asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(n2) : "v"((uint32_t)(p0 >> 32)), "v"(c3), "s"(k1));
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void rng_block(const TraceParams& P, uint32_t pixel, uint32_t sample, uint32_t event, uint32_t block, float out[4])
{
    uint32_t r[4];
    philox4x32_10(event * 32u + block, sample, pixel, 0u, P.key0, P.key1, r);
    out[0] = u32_to_unit(r[0]); out[1] = u32_to_unit(r[1]); out[2] = u32_to_unit(r[2]); out[3] = u32_to_unit(r[3]);
}

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t rank_in(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// sqrtf(x), correctly rounded, in 9 instructions instead of the compiler's 21.  The compiler's expansion is v_sqrt_f32 (1 ulp), the two
// neighbours y -/+ 1 ulp, two fused residuals that pick among the three -- and, around that, a rescaling of arguments below 2^-96 (whose
// residuals would underflow) and a class test for 0 / inf.  This is the same selection without the wrapping; when any lane's argument is
// outside [2^-96, inf) (a NaN included) the whole wave takes the compiler's sqrtf.  Identical results by construction; the device checks
// it for every one of the 2^32 bit patterns (rtx_debug_quotient_check mode 3, tests/test_gpu_edges.py).
__device__ __forceinline__ float sqrt_rn(float x)
{
    if (__ballot(!(x >= 1.262177448e-29f && x < __builtin_inff()))) return sqrtf(x);
    const float y  = __builtin_amdgcn_sqrtf(x);
    const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    const float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
    float r = (0.0f >= rm) ? ym : y;
    r = (0.0f < rp) ? yp : r;
    return r;
}

// Diagnostic build only (-DRTX_PHASE_TIMERS): per-phase wave cycles, s_memtime deltas summed into stats[kStCount + slot].
#ifdef RTX_PHASE_TIMERS
#define RTX_STAMP(slot) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); ph[slot] += now_ - ph_last; ph_last = now_; } while (0)
#else
#define RTX_STAMP(slot) do { } while (0)
#endif

enum LaneState : uint32_t { kNeedItem = 0, kNeedSample = 1, kTrace = 2, kDead = 3, kWalking = 4 };     // kWalking (grid variant): the ray's walk is parked in LDS and goes on in the next iteration

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));

// Two-way f16 split of an f32 (round to nearest): x = x1 + x2 + e with |e| <= 2^-22 |x| (or 2^-25 once x2 is subnormal).
// Returns x1 in the low and x2 in the high half: the ray (B) side's k-slot pattern (p1, p2, p1, p2) is this dword twice,
// meeting the sphere side's (s1, s1, s2, s2).
__device__ __forceinline__ uint32_t split_f16_pair(float x)
{
    const _Float16 x1 = (_Float16)x;
    const float    r  = x - (float)x1;          // exact
    h2v p;
    p[0] = x1;
    p[1] = (_Float16)r;
    return __builtin_bit_cast(uint32_t, p);
}

// ---------------------------------------------------------------------------------------------------------------
// One 16-byte-per-lane global -> LDS copy (1 KB per wave-instruction, no VGPR destination): lane l's 16 bytes land at
// lds_dst + 16 l.  The compiler does not count it: wait with s_waitcnt vmcnt(0) before the barrier that publishes the data.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#ifndef RTX_GRID_WAVES
#define RTX_GRID_WAVES 6
#endif
template <int FILTER, int TABLE, bool VALIDATE>
__global__ void __launch_bounds__(kMaxBlock) __attribute__((amdgpu_waves_per_eu(FILTER == kFilterGrid ? RTX_GRID_WAVES : 4))) trace_kernel(const TraceParams P)   // <= 128 VGPRs; the grid walk hides its dependent loads better with 6 waves of <= 80 (36.6 -> 32.3 ms at 20 164 spheres; 8 waves spill)
{
    constexpr bool LDS_TABLE = (TABLE == kTableLds);
    constexpr bool STREAM    = (TABLE == kTableStream);
    static_assert(!STREAM || FILTER == kFilterMfma, "only the MFMA filter streams its table");
    extern __shared__ float4 smem[];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    // (1) stage the filter's sphere table into LDS once per workgroup (16 B per sphere for the VALU filter,
    //     64 B per sphere -- the f16 A operands of 32-sphere tiles, two MFMAs each, plus one spare tile -- for the MFMA filter)
    const float4* table = (FILTER == kFilterMfma) ? P.hot_mfma : P.hot;
    const uint32_t table_f4 = (FILTER == kFilterMfma) ? (P.n_pad_mfma + 32u) * 4u : (FILTER == kFilterGrid ? 0u : P.n_pad);
    // sample-chunk boundaries of this launch (TraceParams::chunk_tab), kChunkTabBytes at the very front of the LDS
    uint32_t* const ctab = reinterpret_cast<uint32_t*>(smem);
    if (P.chunk_spp == 0u)
        for (uint32_t i = threadIdx.x; i <= P.nchunks; i += blockDim.x) ctab[i] = P.chunk_tab[i];
    float4* const lds0 = smem + kChunkTabBytes / 16u;
    float4* wave_lds;
    const float4* exact = P.exact;
    if (LDS_TABLE) {
        // table, then the exact (cx, cy, cz, r) records the candidate re-test reads, then the per-wave areas
        for (uint32_t i = threadIdx.x; i < table_f4; i += blockDim.x) lds0[i] = table[i];
        for (uint32_t i = threadIdx.x; i < P.n; i += blockDim.x) lds0[table_f4 + i] = P.exact[i];
        table = lds0;
        exact = lds0 + table_f4;
        wave_lds = lds0 + table_f4 + ((P.n + 3u) & ~3u);
    } else if (STREAM) {
        wave_lds = lds0 + kStreamStageBytes / 16u;      // the stage comes first
    } else {
        wave_lds = lds0;
    }
    __syncthreads();
    // per wave (kWaveLdsBytes): 64 ray records in two planes, (o, a) and (d, -), that the candidate re-test gathers from -- the MFMA filter
    // first uses the same 2 KB to hand its per-ray operands around --, 64 result slots, and the candidate queue
    wave_lds += wave * (wave_lds_bytes(FILTER) / 16u);
    float4* const             rayrec = wave_lds;
    float* const              xbuf   = reinterpret_cast<float*>(wave_lds);
    unsigned long long* const res    = reinterpret_cast<unsigned long long*>(wave_lds + 128);
    uint32_t* const           queue  = reinterpret_cast<uint32_t*>(wave_lds + 160);   // 256 entries
    uint2* const              wordq  = reinterpret_cast<uint2*>(wave_lds + 224);      // 128 entries (MFMA filter)

    // The kernel's parameters live in the kernel-argument segment.  Held in scalar registers for the whole kernel they do not fit: the
    // compiler spills them to vector lanes and every use becomes a v_readlane -- a vector instruction in a loop that is bound by the vector
    // instructions it issues.  So the phases that run once per iteration read what they need from the segment when they run (scalar loads,
    // served by the scalar cache); the empty asm keeps the compiler from hoisting the loads out of the loop again.
    auto kargs = [&]() {
        const char __attribute__((address_space(4)))* k_ = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(k_));
        return (const TraceParams __attribute__((address_space(4)))*)k_;
    };
    // wave-level work reservation: [w_next, w_end) is owned, pf holds the next reservation (lane 0's value)
    uint32_t pf = 0;
    if (lane == 0) pf = atomicAdd(P.work_counter, P.wave_batch);
    uint32_t w_next = uniform(pf);
    uint32_t w_end  = w_next + P.wave_batch;
    // The next reservation is fetched ahead of time so that the counter's latency stays hidden -- except near the end of the
    // list (the last P.tail_guard items): a wave hands a reserved item out only when one of ITS lanes falls free, so every
    // item held in reserve there can end up waiting for a busy wave while other waves idle; the last reservations are
    // therefore made on demand.
    bool have_pf = w_end + P.tail_guard < P.n_items;
    if (have_pf && lane == 0) pf = atomicAdd(P.work_counter, P.wave_batch);
    // Cost-ordered launches (P.order != NULL): the work list is a list of batches of 64 items = one STRIP of 64 consecutive
    // pixels x one sample chunk, sorted by the host by expected duration, longest first (order[b] = strip | chunk << 22), and
    // a reservation is one batch.  Which strip and chunk a batch is, is wave-uniform (scalar registers, one scalar load per
    // batch); a lane's item is the batch's pixel base + its offset.
    uint32_t cur_base = 0u, cur_chunk = 0u;
    auto batch_info = [&](const TraceParams __attribute__((address_space(4)))* Q, uint32_t start, uint32_t& base, uint32_t& chunk) {
        const uint32_t e = start < Q->n_items ? Q->order[start >> 6] : 0xFFFFFFFFu;
        chunk = e >> 22;
        base  = e == 0xFFFFFFFFu ? 0xFFFFFFC0u : ((e & 0x3FFFFFu) << 6);    // past the end: no pixel of the batch is valid
    };
    if (P.order) batch_info(kargs(), w_next, cur_base, cur_chunk);

    // per-lane path state
    uint32_t st = kNeedItem, item = 0, pix = 0, prow = 0, s_cur = 0, s_end = 0, bounce = 0;
    uint32_t item_rays = 0;      // _sceneHit calls of the current item: the pixel's cost, stored beside its partial sum
    V3       o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(1, 1, 1), acc = mk(0, 0, 0);
    // wave-level counters
    uint32_t c_samples = 0, c_miss = 0, c_absorb = 0, c_depth = 0, c_filter_miss = 0, c_cand = 0;
    uint32_t lane_tests = 0;     // grid filter: exact tests this lane ran (the other filters count candidates per batch, wave-wide)
    uint64_t c_rays = 0;

    // -r only: per-lane stack of attenuations in global memory, entry k of this lane at att_stack[k * stack_stride]
    const bool     recursive    = (P.flags & RTX_FLAG_RECURSIVE) != 0u;
    const size_t   stack_stride = (size_t)gridDim.x * blockDim.x;
    float4* const  att_stack    = P.att_stack + ((size_t)blockIdx.x * blockDim.x + threadIdx.x);

    // The camera's 19 constants: scalar registers for the whole kernel -- except in the grid variant, whose walk loop needs the
    // scalar registers for itself (spilled ones come back as v_readlane, i.e. vector instructions, on every trip): there phase (B)
    // reads them from the kernel-argument segment when it runs (the empty asm keeps the loads from being hoisted again).
    V3 cam_o = ld3(P.cam.origin), cam_lc = ld3(P.cam.left_corner), cam_h = ld3(P.cam.horizontal), cam_v = ld3(P.cam.vertical);
    V3 cam_uu = ld3(P.cam.u), cam_vv = ld3(P.cam.v);
    float cam_lens = P.cam.lens_radius;

#ifdef RTX_PHASE_TIMERS
    uint64_t ph[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    uint64_t ph_last = __builtin_amdgcn_s_memtime();
#endif
#ifdef RTX_WALK_STATS
    // diagnostic build (grid walk): lane k keeps counter k of P.walk_stats[0..11] (ws_acc) and of the four 9-bucket histograms at [16..51] (ws_hist)
    uint32_t ws_acc = 0u, ws_hist = 0u, ws_lin = 0u;      // ws_lin: lane k counts the rays that needed k loop trips (63: or more), P.walk_stats[64..127]
#define WS_ADD(slot, v) do { if (lane == (uint32_t)(slot)) ws_acc += (uint32_t)(v); } while (0)
#define WS_HIST(h, n, active) do { const uint32_t bk_ = (n) == 0u ? 0u : (32u - (uint32_t)__clz((int)(n)) > 8u ? 8u : 32u - (uint32_t)__clz((int)(n))); \
        for (uint32_t b_ = 0; b_ < 9u; ++b_) { const uint32_t c_ = (uint32_t)__popcll(__ballot((active) && bk_ == b_)); if (lane == 9u * (h) + b_) ws_hist += c_; } } while (0)
#endif
#ifdef RTX_WAVE_TIMES
    // diagnostic build: per wave (start, counter seen dry, end) on the chip-wide 100 MHz clock + iterations + lane-iterations spent tracing
    const uint64_t wt_start = __builtin_amdgcn_s_memrealtime();
    uint64_t wt_dry = 0, wt_iters = 0, wt_lanes = 0;
    uint32_t wt_item_t0 = 0, wt_last_t0 = 0, wt_last_rays = 0, wt_last_pix = 0, wt_last_len = 0;     // the item that finished last in this wave
#endif
    for (;;) {
#ifdef RTX_WAVE_TIMES
        if (wt_dry == 0 && w_next >= P.n_items) wt_dry = __builtin_amdgcn_s_memrealtime();
        wt_iters++;
#endif
        // ---- (A) lanes whose item is finished take the next one (persistent tile fetch) -------------------------
        uint64_t need = __ballot(st == kNeedItem);
        if (need) {
            const TraceParams __attribute__((address_space(4)))* const PA = kargs();
            uint32_t cnt = (uint32_t)__popcll(need), r = rank_in(need), avail = w_end - w_next;
            uint32_t chunk = 0u, npx = 0xFFFFFFFFu;      // the lane's new item: (chunk, pixel); pixel >= npix: none
            bool     dry = false;
            if (PA->order) {
                // strip batches of 64: the lane's pixel is its batch's base + its offset in the batch
                uint32_t base = cur_base, off = (w_next + 64u - w_end) + r;
                chunk = cur_chunk;
                if (cnt <= avail) {
                    w_next += cnt;
                } else {
                    if (!have_pf && lane == 0) pf = atomicAdd(PA->work_counter, 64u);
                    const uint32_t nb = uniform(pf);
                    uint32_t nbase, nchunk;
                    batch_info(PA, nb, nbase, nchunk);
                    if (r >= avail) { base = nbase; chunk = nchunk; off = r - avail; }
                    w_next = nb + (cnt - avail);
                    w_end  = nb + 64u;
                    cur_base = nbase; cur_chunk = nchunk;
                    have_pf = w_end + PA->tail_guard < PA->n_items;
                    if (have_pf && lane == 0) pf = atomicAdd(PA->work_counter, 64u);
                }
                npx = base + off;
                dry = base == 0xFFFFFFC0u;
            } else {
                uint32_t it;
                if (cnt <= avail) {
                    it = w_next + r;
                    w_next += cnt;
                } else {
                    if (!have_pf && lane == 0) pf = atomicAdd(PA->work_counter, PA->wave_batch);
                    uint32_t nb = uniform(pf);
                    it     = (r < avail) ? (w_next + r) : (nb + (r - avail));
                    w_next = nb + (cnt - avail);
                    w_end  = nb + PA->wave_batch;
                    have_pf = w_end + PA->tail_guard < PA->n_items;
                    if (have_pf && lane == 0) pf = atomicAdd(PA->work_counter, PA->wave_batch);
                }
                if (it < PA->n_items) {
                    chunk = it / PA->npix;
                    npx   = it - chunk * PA->npix;
                } else {
                    dry = true;
                }
            }
            if (st == kNeedItem) {
                if (dry) {
                    st = kDead;
                } else if (npx < PA->npix) {                 // (the last strip of an ordered list may reach past the image: such a lane asks again)
                    pix            = npx;
                    item           = chunk * PA->npix + pix;
                    item_rays      = 0u;
                    prow           = pix / PA->cols;                     // once per item, not per sample
                    if (PA->chunk_spp) {                                    // uniform chunks (caller's choice)
                        s_cur          = PA->spp_begin + chunk * PA->chunk_spp;
                        uint32_t e     = s_cur + PA->chunk_spp, lim = PA->spp_begin + PA->spp_count;
                        s_end          = e < lim ? e : lim;
                    } else {                                              // the launch's chunk plan
                        s_cur = PA->spp_begin + ctab[chunk];
                        s_end = PA->spp_begin + ctab[chunk + 1u];
                    }
                    acc            = mk(0, 0, 0);
                    st             = kNeedSample;
#ifdef RTX_WAVE_TIMES
                    wt_item_t0 = (uint32_t)(__builtin_amdgcn_s_memrealtime() - wt_start);
#endif
                }
            }
        }

        RTX_STAMP(0);
        // ---- (B) camera rays for lanes starting a sample: raytracer.cpp:177-179 + camera.h:75-86 ------------------
        if (__ballot(st == kNeedSample)) {
            const bool go = (st == kNeedSample);
            const uint32_t px = pix - prow * P.cols, py = prow;
            float      b[4];
            rng_block(P, pix, s_cur, 0, 0, b);
            // (x + jitter) / cols as (float)((double)numerator * (1.0 / (double)cols)): the identity of `quot` below (a float quotient through a
            // double reciprocal, checked on the device); numerator and result are zero or normal here, the reciprocals come from the host
            float u = (float)((double)((float)px + b[0]) * P.inv_cols);
            float v = (float)((double)((float)py + b[1]) * P.inv_rows);
            // randomOnUnitDisk, utils.cu:85-94: try 0 from words (2,3) of block 0, try t>=1 from block (t+1)/2
            float    dkx = 2.0f * b[2] - 1.0f, dky = 2.0f * b[3] - 1.0f;
            bool     pending = go && (((dkx * dkx) + (dky * dky)) + 0.0f >= 1.0f);
            uint32_t t = 1;
            while (__ballot(pending)) {          // t <= 20: at most 21 tries in all
                float bb[4];
                rng_block(P, pix, s_cur, 0, (t + 1) >> 1, bb);
                if (pending) {
                    dkx = 2.0f * bb[0] - 1.0f; dky = 2.0f * bb[1] - 1.0f;
                    if (!((((dkx * dkx) + (dky * dky)) + 0.0f >= 1.0f) && t < 20)) pending = false;
                }
                t++;
                if (pending) {
                    dkx = 2.0f * bb[2] - 1.0f; dky = 2.0f * bb[3] - 1.0f;
                    if (!((((dkx * dkx) + (dky * dky)) + 0.0f >= 1.0f) && t < 20)) pending = false;
                }
                t++;
            }
            if (go) {
                if constexpr (FILTER == kFilterGrid) {
                    const char __attribute__((address_space(4)))* kargs = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(kargs));
                    const rtx_camera __attribute__((address_space(4)))* cam = (const rtx_camera __attribute__((address_space(4)))*)(kargs + offsetof(TraceParams, cam));
                    cam_o  = mk(cam->origin[0], cam->origin[1], cam->origin[2]);       cam_lc = mk(cam->left_corner[0], cam->left_corner[1], cam->left_corner[2]);
                    cam_h  = mk(cam->horizontal[0], cam->horizontal[1], cam->horizontal[2]); cam_v  = mk(cam->vertical[0], cam->vertical[1], cam->vertical[2]);
                    cam_uu = mk(cam->u[0], cam->u[1], cam->u[2]);                       cam_vv = mk(cam->v[0], cam->v[1], cam->v[2]);
                    cam_lens = cam->lens_radius;
                }
                float rx = cam_lens * dkx, ry = cam_lens * dky;
                V3    offset = add(scale(rx, cam_uu), scale(ry, cam_vv));
                o = add(cam_o, offset);
                d = sub(sub(add(add(cam_lc, scale(u, cam_h)), scale(1.0f - v, cam_v)), cam_o), offset);
                thr    = mk(1, 1, 1);
                bounce = 0;
                st     = kTrace;
            }
        }

        RTX_STAMP(1);
        const bool     tracing  = (st == kTrace);                                    // rays that start their _sceneHit in this iteration
        const uint64_t tmask    = __ballot(tracing);
        const uint64_t amask    = (FILTER == kFilterGrid) ? __ballot(tracing || st == kWalking) : tmask;     // grid variant: + the rays whose walk was parked in an earlier iteration
        if (STREAM) {
            // the waves of a workgroup walk the table together (barriers in the filter loop), so they also stop together;
            // this barrier also separates the previous iteration's reads of the stage from the next fill
            if (!__syncthreads_or(tmask != 0ull)) break;
        } else {
            if (!amask) break;      // every lane is kDead: the queue is drained and all paths have ended
        }
        c_rays += (uint64_t)__popcll(tmask);
        item_rays += tracing ? 1u : 0u;
#ifdef RTX_WAVE_TIMES
        wt_lanes += (uint64_t)__popcll(tmask);
#endif

        // ---- (C) _sceneHit: conservative filter over all spheres, exact re-test of the candidates ----------------
        const float a = dot(d, d);                                        // sphere.cpp:73
        float       closest = 3.402823466e+38f;                           // FLT_MAX, raytracer.cpp:251
        uint32_t    best = kNoHit;
        if constexpr (FILTER == kFilterGrid) {
            // ---- Uniform grid (host side and the argument why no hit is lost: rtx_host.cpp, build_grid).
            // Each lane walks the cells that its own ray crosses, in order, and runs the exact test of sphere.cpp:72-99 on the
            // spheres registered there; the big list (spheres too large for the grid: the ground) is tested first, for every ray.
            // The winner is the minimum of (t, index) as everywhere else, so the order in which spheres are met does not matter,
            // and a sphere met twice (it is registered in every cell it overlaps) changes nothing.  Registrations are grown by
            // the blur of the reference's fp32 discriminant, in tiers by distance from the ray's origin; the walk stops early
            // only when the closest hit so far lies more than twice the walk's own slack before the exit of the cell just
            // finished (the accepted root of an untested sphere lies on its registered stretch of the ray, which starts later).
            // sphere.cpp:72-99 on geometry s; the sphere's index is only needed when the hit could become the closest one
#ifdef RTX_WALK_STATS
            uint32_t ws_roots = 0u;
#endif
            // The roots' two divisions by a = d.d: x / a == (float)((double)x * (1.0 / (double)a)) bit for bit -- the exact quotient of two
            // floats is either a float or at least 2^-49 (relative) away from every rounding boundary of the float format (the boundary's 25-bit
            // significand times a's 24 bits cannot miss x's by less than one unit), and the double product is within 1.5 * 2^-52 of it; results
            // that round into the subnormal range may differ, but a root is only ever compared with 0.001 and FLT_MAX.  One double reciprocal
            // per ray, then convert - multiply - convert per root instead of the 12-instruction IEEE sequence (tests/test_gpu_edges.py checks
            // the identity on the device over 2^32 operand pairs).
            const double rd = 1.0 / (double)a;
            auto quot = [&](float x) { return (float)((double)x * rd); };
            // (written with selects: the walk loop is bound by the instructions it issues, scalar ones -- the lane masks of nested
            // conditions -- included; the one branch left skips the roots when no lane's discriminant is positive)
            auto test = [&](const bool active, const float4 s, const uint32_t i) {
                lane_tests += active ? 1u : 0u;
                const V3    oc = sub(o, mk(s.x, s.y, s.z));
                const float bq = dot(oc, d);
                const float cq = dot(oc, oc) - (s.w * s.w);
                const float disc = bq * bq - a * cq;
                if (active && disc > 0) {
#ifdef RTX_WALK_STATS
                    ws_roots++;
#endif
                    const float sq = sqrt_rn(disc);
                    float       t  = quot(-bq - sq);
                    if (!(t > 0.001f)) t = quot(-bq + sq);
                    const bool take = t > 0.001f && t < 3.402823466e+38f && (t < closest || (t == closest && i < best));
                    closest = take ? t : closest;
                    best    = take ? i : best;
                }
            };
            for (uint32_t k = 0; k < P.grid.n_big; ++k) {
                const uint32_t i = P.grid_big[k];
                test(tracing, P.exact[i], i);
            }
            RTX_STAMP(8);      // (grid: the big list)
            bool scan_all = false;
            // the grid's 24 constants are read from the kernel-argument segment here, in every iteration, like the camera's in phase (B): kept in
            // scalar registers for the whole kernel they are spilled to vector lanes and come back as v_readlane in every set-up
            const char __attribute__((address_space(4)))* gargs = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(gargs));
            const GridDesc __attribute__((address_space(4)))* const G = (const GridDesc __attribute__((address_space(4)))*)(gargs + offsetof(TraceParams, grid));
            if (G->dim[0] != 0u) {
                const float inf = __builtin_inff();
                const float slack_t = 2.0f * G->pad * __builtin_amdgcn_rsqf(a);        // twice the walk's slack, as a ray parameter
                // Is the origin within the registrations' reach?  If some binned sphere can be farther away than the outermost tier
                // allows, the ray is clipped against the box grown by the blur of that distance and, if it enters, scans the table.
                const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
                float s_near2 = 0.0f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float lo = G->gmin[k], hi = lo + (float)G->dim[k] * G->h[k];
                    const float off = fmaxf(fmaxf(lo - oo[k], oo[k] - hi), 0.0f);
                    s_near2 += off * off;
                }
                // (the walk only needs conservative values here: hardware sqrt / rcp, 1 ulp, with margins of 1e-5 -- its results do not depend on them)
                const float l_far = __builtin_amdgcn_sqrtf(s_near2) * 1.00001f + G->box_diag;
                const bool  far   = !(l_far <= G->far_l);                               // also NaN
                const float rm    = G->r_min;
                const float extra = far ? G->pad + (__builtin_amdgcn_sqrtf(rm * rm + 1.9073486328125e-6f * (l_far * l_far + rm * rm)) * 1.00001f - rm) : 0.0f;
                // slab clip of the ray against the grid box; a zero component of d leaves that axis unbounded when the origin is inside
                float    t_in = 0.0f, t_out = inf;
                bool     walking = tracing;
                float    inv[3];
                bool     moving[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float lo = G->gmin[k] - extra, hi = G->gmin[k] + (float)G->dim[k] * G->h[k] + extra;
                    inv[k] = __builtin_amdgcn_rcpf(dd[k]);          // off by an ulp: cell-exit parameters move by 1e-7 of themselves, the slack is 2^-12 of the extent
                    moving[k] = __builtin_fabsf(inv[k]) < inf;      // a zero (or denormal: v_rcp gives inf) component does not move the ray along this axis
                    if (moving[k]) {
                        const float ta = (lo - oo[k]) * inv[k], tb = (hi - oo[k]) * inv[k];
                        t_in  = fmaxf(t_in, fminf(ta, tb));
                        t_out = fminf(t_out, fmaxf(ta, tb));
                    } else if (oo[k] < lo || oo[k] > hi) {
                        walking = false;
                    }
                }
                walking = walking && t_in <= t_out;
                if (walking && far) { scan_all = true; walking = false; }
                // Per axis: tn = parameter at which the ray leaves its current cell, dt = parameter per cell, the signed step of the
                // linear cell index, and (10 bits each, packed in `rem`) how many cells are left before the ray leaves the grid.
                // tn accumulates one rounding per step of its axis: at most 255 x 2^-24 of a parameter whose distance equivalent is
                // at most far_l (two diagonals), i.e. under a quarter of `pad` (2^-12 of the extent); the early stop leaves 2 pad.
                float    tn[3], dt[3];
                int      cstride[3];
                uint32_t rem = 0u, cidx = 0u;
                const uint32_t stride[3] = { 1u, G->dim[0], G->dim[0] * G->dim[1] };
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float p = oo[k] + t_in * dd[k];
                    int c = (int)floorf((p - G->gmin[k]) * G->inv_h[k]);
                    c = c < 0 ? 0 : (c >= (int)G->dim[k] ? (int)G->dim[k] - 1 : c);
                    cidx += (uint32_t)c * stride[k];
                    const bool fwd = dd[k] > 0.0f;
                    if (moving[k]) {
                        tn[k] = (G->gmin[k] + (float)(c + (fwd ? 1 : 0)) * G->h[k] - oo[k]) * inv[k];
                        dt[k] = G->h[k] * __builtin_fabsf(inv[k]);
                        rem  |= (uint32_t)(fwd ? (int)G->dim[k] - 1 - c : c) << (10 * k);
                    } else {
                        tn[k] = inf;
                        dt[k] = 0.0f;
                    }
                    cstride[k] = fwd ? (int)stride[k] : -(int)stride[k];
                }
                // Registration tier of this ray: the lists that are valid up to the farthest point the walk can reach, i.e. the
                // box's exit or the hit already found on the big list (the ground), whichever is nearer -- one decision per ray
                // instead of one per cell (the outermost tier is valid wherever the origin passed the test above).
                const float s_max = fminf(t_out, closest + slack_t) * (__builtin_amdgcn_sqrtf(a) * 1.00001f);     // as a distance, rounded up
                const bool  tier1_fresh = !(s_max <= G->tier_s[0]), tier2_fresh = !(s_max <= G->tier_s[1]);   // NaN: the widest lists
                uint32_t it = 0u, it_end = 0u;
                bool     fresh_cell = true;
                // Lane decoupling: a wave's walk lasts as long as its longest ray's, and most of its trips run for a
                // handful of rays.  When at most P.walk_park rays are still walking while at least half the wave waits to be shaded, their
                // state is PARKED in LDS (13 words per lane, wave_lds + 2304 bytes: behind what phase (D)'s cooperative sampling uses), the
                // wave shades the finished lanes and fetches their next rays, and the parked walks go on beside those in the next
                // iteration.  o and d of a parked lane stay in its registers: (D) and (E) do not touch it.
                float* const park = reinterpret_cast<float*>(wave_lds) + 576u + lane;         // word k of this lane at park[64 k]
                bool tier1 = tier1_fresh, tier2 = tier2_fresh;
                if (amask != tmask) {
                    if (st == kWalking) {
                        tn[0] = park[0];   tn[1] = park[64];  tn[2] = park[128];
                        dt[0] = park[192]; dt[1] = park[256]; dt[2] = park[320];
                        rem     = __float_as_uint(park[384]);
                        cidx    = __float_as_uint(park[448]);
                        it      = __float_as_uint(park[512]);
                        it_end  = __float_as_uint(park[576]);
                        closest = park[640];
                        best    = __float_as_uint(park[704]);
                        const uint32_t fl = __float_as_uint(park[768]);
                        tier1 = (fl & 1u) != 0u; tier2 = (fl & 2u) != 0u; fresh_cell = (fl & 4u) != 0u;
                        walking = true;
                    }
                }
                // (no parking while the wave drains at the end of the launch: the lanes it would free have nothing to fetch)
                const uint32_t park_lim = (uint32_t)__popcll(amask) >= 48u ? P.walk_park : 0u;
                RTX_STAMP(9);  // (grid: clip, cell and tier set-up)
#ifdef RTX_WALK_STATS
                uint32_t ws_t = 0u, l_trips = 0u, l_cells = 0u, l_tests = 0u;
                ws_roots = 0u;
#endif
                if (__ballot(walking)) for (;;) {
#ifdef RTX_WALK_STATS
                    ws_t++; WS_ADD(2, 1); { const uint64_t m_ = __ballot(walking); WS_ADD(3, __popcll(m_)); }
                    if (walking) l_trips++;
                    { const uint64_t m_ = __ballot(walking && fresh_cell); if (m_) { WS_ADD(4, 1); WS_ADD(5, __popcll(m_)); } if (walking && fresh_cell) l_cells++; }
                    const uint32_t ws_r0 = ws_roots;
#endif
                    // One fetch of 32 bytes per walking ray and trip, one wait: the record of the cell just entered -- (first entry, its sphere's
                    // index, entries per tier, that sphere's geometry) -- or the next entry of the cell the ray is in (geometry, index).
                    float4 ra, rb;
                    if (walking) {
                        const float4* const p = fresh_cell ? P.grid_cells + 2u * (size_t)cidx : P.grid_recs + 2u * (size_t)it;
                        ra = p[0];
                        rb = p[1];
                    }
#ifdef RTX_PROBE_VALU      // diagnostic: what do RTX_PROBE_VALU more vector instructions per trip cost?
                    { float pv = ra.x;
#pragma unroll
                      for (int k_ = 0; k_ < RTX_PROBE_VALU; ++k_) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(pv));
                      asm volatile("" :: "v"(pv)); }
#endif
#ifdef RTX_PROBE_SALU      // diagnostic: ... scalar instructions
                    { uint32_t ps = 1u;
#pragma unroll
                      for (int k_ = 0; k_ < RTX_PROBE_SALU; ++k_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ps) :: "scc");
                      asm volatile("" :: "s"(ps)); }
#endif
#ifdef RTX_PROBE_LOAD      // diagnostic: ... one more dependent fetch
                    if (walking) { const float4 x_ = P.grid_cells[__float_as_uint(ra.x) & 1u]; asm volatile("s_waitcnt vmcnt(0)" :: "v"(x_.x) : "memory"); }
#endif
                    const uint32_t cnts = __float_as_uint(ra.z);
                    const uint32_t cnt  = tier2 ? __float_as_uint(ra.w) : (tier1 ? cnts >> 16 : (cnts & 0xFFFFu));
                    const uint32_t idx  = __float_as_uint(fresh_cell ? ra.y : rb.x);
                    const float4   geo  = fresh_cell ? rb : ra;
                    const bool     entered = walking && fresh_cell;
                    it     = entered ? __float_as_uint(ra.x) : it;
                    it_end = entered ? it + cnt : it_end;
                    const bool has = walking && it < it_end;
#ifdef RTX_WALK_STATS
                    { const uint64_t m_ = __ballot(has); if (m_) { WS_ADD(6, 1); WS_ADD(7, __popcll(m_)); } if (has) l_tests++; }
#endif
                    test(has, geo, idx);
                    it += has ? 1u : 0u;
#ifdef RTX_WALK_STATS
                    { const uint64_t m_ = __ballot(ws_roots != ws_r0); if (m_) { WS_ADD(10, 1); WS_ADD(11, __popcll(m_)); } }
                    { const uint64_t m_ = __ballot(walking && it >= it_end); if (m_) { WS_ADD(8, 1); WS_ADD(9, __popcll(m_)); } }
#endif
                    // a ray whose cell is done stops if its closest hit lies safely before the cell's exit (or the box ends), else steps to the next cell
                    {
                        const bool  done = walking && it >= it_end;
                        const float t_leave = fminf(tn[0], fminf(tn[1], tn[2]));
                        const bool  ax0 = tn[0] <= tn[1] && tn[0] <= tn[2], ax1 = !ax0 && tn[1] <= tn[2];
                        const uint32_t sh = ax0 ? 0u : (ax1 ? 10u : 20u);
                        const bool  stop = closest + slack_t + closest * 4.0e-6f < t_leave || ((rem >> sh) & 1023u) == 0u;
                        const bool  adv  = done && !stop;
                        walking = walking && !(done && stop);
                        rem  -= adv ? (1u << sh) : 0u;
                        cidx += adv ? (uint32_t)(ax0 ? cstride[0] : (ax1 ? cstride[1] : cstride[2])) : 0u;
                        tn[0] += (adv && ax0) ? dt[0] : 0.0f;
                        tn[1] += (adv && ax1) ? dt[1] : 0.0f;
                        tn[2] += (adv && !ax0 && !ax1) ? dt[2] : 0.0f;
                        fresh_cell = adv;
                    }
                    // every trip has served every walking ray (parked ones included: they advance in each iteration); go on while enough of them are left
                    if ((uint32_t)__popcll(__ballot(walking)) <= park_lim) break;
                }
                if (__ballot(walking)) {
                    if (walking) {          // parked: everything the loop carries (the strides follow from d, the slack from a)
                        park[0]   = tn[0]; park[64]  = tn[1]; park[128] = tn[2];
                        park[192] = dt[0]; park[256] = dt[1]; park[320] = dt[2];
                        park[384] = __uint_as_float(rem);
                        park[448] = __uint_as_float(cidx);
                        park[512] = __uint_as_float(it);
                        park[576] = __uint_as_float(it_end);
                        park[640] = closest;
                        park[704] = __uint_as_float(best);
                        park[768] = __uint_as_float((tier1 ? 1u : 0u) | (tier2 ? 2u : 0u) | (fresh_cell ? 4u : 0u));
                        st = kWalking;
                    }
                }
                if (!walking && st == kWalking) st = kTrace;       // a parked walk has ended: its ray is shaded with the others
                RTX_STAMP(2);  // (grid: the walk)
#ifdef RTX_WALK_STATS
                WS_ADD(0, 1); WS_ADD(1, __popcll(tmask));
                WS_HIST(0u, ws_t, lane == 0u); WS_HIST(1u, l_trips, tracing); WS_HIST(2u, l_cells, tracing); WS_HIST(3u, l_tests, tracing);
                for (uint32_t b_ = 0; b_ < 64u; ++b_) { const uint32_t c_ = (uint32_t)__popcll(__ballot(tracing && (l_trips < 63u ? l_trips : 63u) == b_)); if (lane == b_) ws_lin += c_; }
#endif
            }
            // rays that outran the grid's registrations (origins many scene diameters away): the reference's own loop
            if (__ballot(scan_all)) {
                for (uint32_t i = 0; i < P.n; ++i) {
                    const float4 s = P.exact[i];
                    test(scan_all, s, i);
                }
            }
        } else {
            // Candidates of all 64 rays go through one wave-wide queue of (ray lane << 26 | entry) words, so that the
            // exact re-test always runs with full lanes: a batch is 64 queue entries, one per lane, whatever rays they
            // belong to.  The testing lane gathers the ray from its record and folds its result into the ray's slot with
            // a 64-bit LDS minimum on (t bits << 32 | sphere index).
            //
            // sphere.cpp:72-99 for one candidate.  Candidates arrive in any order, so instead of relying on the
            // ascending scan of raytracer.cpp:291-300 the tie rule is explicit: the hit distance of sphere i is its
            // first root beyond t_min (the far root is only ever accepted when the near one is <= t_min, because
            // far >= near), and among equal distances the lowest index wins, exactly as the strict '<' there decides:
            // that is the minimum of (t, i) in lexicographic order, and t > 0 orders like its bit pattern.
            uint32_t q_head = 0, q_count = 0;          // wave-uniform
            res[lane] = 0x7F7FFFFFFFFFFFFFull;         // (FLT_MAX, kNoHit): raytracer.cpp:251
            auto run_batch = [&](uint32_t nb) {
                RTX_STAMP(10);
                if (lane < nb) {
                    const uint32_t ent = queue[(q_head + lane) & 255u];
                    const uint32_t ray = ent >> 26;
                    uint32_t       i   = ent & 0x03FFFFFFu;
                    if (FILTER == kFilterMfma) i ^= 31u;     // (group, word, bit) -> sphere index, see the extraction below
                    const float4 r0 = rayrec[ray], r1 = rayrec[64u + ray];
                    const float4 s  = exact[i < P.n ? i : 0u];
                    const V3    ro = mk(r0.x, r0.y, r0.z), rd = mk(r1.x, r1.y, r1.z);
                    const float ra = r0.w;
                    const V3    oc = sub(ro, mk(s.x, s.y, s.z));
                    const float bq = dot(oc, rd);
                    const float cq = dot(oc, oc) - (s.w * s.w);
                    const float disc = bq * bq - ra * cq;
                    if (disc > 0 && i < P.n) {            // i >= n: a padded row, flagged only by a NaN / Inf ray (the reference finds no hit for one)
                        const float sq = sqrt_rn(disc);
                        float       t  = (-bq - sq) / ra;
                        if (!(t > 0.001f)) t = (-bq + sq) / ra;
                        if (t > 0.001f && t < 3.402823466e+38f)    // sphere.cpp:81,90: t < max, and max starts at FLT_MAX
                            __hip_atomic_fetch_min(&res[ray], ((unsigned long long)__float_as_uint(t) << 32) | i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                q_head += nb;
                q_count -= nb;
                c_cand += nb;
                RTX_STAMP(11);
            };
            // the ring holds 256 entries: at most 63 are pending after the drain, one call adds at most 64
            auto enqueue = [&](bool has, uint32_t entry) {
                const uint64_t m = __ballot(has);
                if (m) {
                    while (q_count >= 64u) run_batch(64u);
                    if (has) queue[(q_head + q_count + rank_in(m)) & 255u] = entry;
                    q_count += (uint32_t)__popcll(m);
                }
            };
            // MFMA filter: sign words are sparse (one in eight is non-zero), so they are compacted first: non-zero words
            // go, with their owner and position code, through a second wave-wide ring, and whenever 64 are pending every
            // lane takes one and peels its bits into the candidate queue.
            uint32_t wq_head = 0, wq_count = 0;        // wave-uniform
            auto drain_words = [&](uint32_t nb) {
                uint32_t w = 0u, code = 0u;
                if (lane < nb) { const uint2 e = wordq[(wq_head + lane) & 127u]; w = e.x; code = e.y; }
                wq_head += nb;
                wq_count -= nb;
                while (__ballot(w != 0u)) {
                    enqueue(w != 0u, code | (uint32_t)__builtin_ctz(w | 0x80000000u));
                    w &= w - 1u;
                }
                RTX_STAMP(10);
            };
            auto push_word = [&](uint32_t wv, uint32_t code) {
                const uint64_t m = __ballot(wv != 0u) & tmask;       // lanes without a ray carry stale operands: masked here
                if (m) {
                    if (wq_count >= 64u) drain_words(64u);
                    if (wv != 0u && tracing) wordq[(wq_head + wq_count + rank_in(m)) & 127u] = make_uint2(wv, (lane << 26) | code);
                    wq_count += (uint32_t)__popcll(m);
                }
            };

            if (FILTER == kFilterValu) {
                const float rs = __builtin_amdgcn_rsqf(a) * (1.0f + kEps);
                const float fx = d.x * rs, fy = d.y * rs, fz = d.z * rs;
                rayrec[lane]       = make_float4(o.x, o.y, o.z, a);            // two planes of 16-byte records: a wave's stores and its
                rayrec[64u + lane] = make_float4(d.x, d.y, d.z, 0.0f);         // in-order reads are then free of bank conflicts (32-byte records: 2-way)
                for (uint32_t base = 0; base < P.n_pad; base += 32u) {
                    uint32_t m = 0;
#pragma unroll
                    for (uint32_t j = 0; j < 32u; ++j) {
                        const float4 s   = table[base + j];
                        const float  ocx = s.x - o.x, ocy = s.y - o.y, ocz = s.z - o.z;
                        const float  h   = __builtin_fmaf(ocz, fz, __builtin_fmaf(ocy, fy, ocx * fx));
                        const float  q   = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, s.w)));
                        const float  nd  = __builtin_fmaf(-h, h, q);     // < 0  <=>  inflated discriminant > 0
                        m = __builtin_amdgcn_alignbit(m, __float_as_uint(nd), 31u);
                    }
                    if (!tracing) m = 0;
                    while (__ballot(m != 0)) {
                        const uint32_t j = (uint32_t)__clz((int)(m | 1u));
                        enqueue(m != 0, (lane << 26) | (base + j));
                        m &= ~(0x80000000u >> j);
                    }
                }
            } else {
                // ---- MFMA filter.  The inflated discriminant test is ONE quadratic form in the sphere,
                //   w = K'' + (|o|^2' - (o.d')^2) + sum_i 2 C_i (o.d' d'_i - o_i) - sum_i C_i^2 d'_i^2 - 2 sum_{i<j} C_i C_j d'_i d'_j
                // (= |oc|^2 - r^2 - (oc.d')^2 minus the margins), i.e. a dot product of ten per-sphere monomials
                // (C_i C_j, C_i, 1) with ten per-ray coefficients: K = 32 = two v_mfma_f32_32x32x16_f16 per 32 spheres x 32
                // rays, and nothing but the sign capture is left for the VALU.  Every factor is split into two f16 parts
                // (x = x1 + x2 + e, |e| <= 2^-22 |x|) and a monomial takes three k-slots, x1 p1 + x1 p2 + x2 p1; K'' (against the
                // exact 2^15) and the per-ray constant (against the sphere side's 2^15) take two.  The sphere side is
                // pre-scaled (sg = P.mfma_scale, a power of two, and 2^-15 on the quadratic monomials) so that everything
                // stays inside the f16 range.  Lane (hf = lane >> 5, c = lane & 31) feeds k-slots 8 hf .. 8 hf + 7 of both
                // MFMAs for ray c (first pair) and ray 32 + c (second pair) and receives, for each of the two rays, the
                // results of spheres 8 b + 4 hf + j (v = 4 b + j) of the tile.  Slot groups (MFMA m, half hf):
                //   (0,0) CxCx, CyCy, const    (0,1) CzCz, CxCy, K''    (1,0) CxCz, CyCz, Cz[x1p1, x1p2]    (1,1) Cx, Cy, Cz[x2p1]
                // Each ray's owner first publishes (d', o.d', o, |o|^2') for the lanes that build its operands.
                const float rs = __builtin_amdgcn_rsqf(a);
                const float fx = d.x * rs, fy = d.y * rs, fz = d.z * rs;
                const float od = __builtin_fmaf(o.z, fz, __builtin_fmaf(o.y, fy, o.x * fx));
                const float oo = __builtin_fmaf(o.z, o.z, __builtin_fmaf(o.y, o.y, o.x * o.x));
                float4* const xbuf4 = reinterpret_cast<float4*>(xbuf);
                xbuf4[lane]       = make_float4(fx, fy, fz, od);
                xbuf4[64u + lane] = make_float4(o.x, o.y, o.z, oo - kGammaRay * oo);      // |o|^2 minus the ray's share of the margin
                const uint32_t hf = lane >> 5, cl = lane & 31u;
                const float    sg = P.mfma_scale, T = 32768.0f;
                h8v b0[2], b1[2];
#pragma unroll
                for (uint32_t half = 0; half < 2u; ++half) {
                    const uint32_t ray = half * 32u + cl;
                    const float4   dv = xbuf4[ray], ov = xbuf4[64u + ray];
                    const float    gz = sg * __builtin_fmaf(dv.w, dv.z, -ov.z);             // sg (o.d' d'_z - o_z), against 2 sg C_z
                    // first MFMA: hf 0: -2^15 d'x^2, -2^15 d'y^2, constant sg^2 2^-15 (|o|^2' - (o.d')^2);  hf 1: -2^15 d'z^2, -2^16 d'x d'y, 2^15
                    const float x0 = -T * (hf ? dv.z * dv.z : dv.x * dv.x);
                    const float y0 = -T * (hf ? 2.0f * (dv.x * dv.y) : dv.y * dv.y);
                    const float e0 = (sg * sg * (1.0f / 32768.0f)) * __builtin_fmaf(-dv.w, dv.w, ov.w);
                    // second MFMA: hf 0: -2^16 d'x d'z, -2^16 d'y d'z, g_z (x1p1, x1p2);  hf 1: g_x, g_y, g_z (x2p1)
                    const float x1 = hf ? sg * __builtin_fmaf(dv.w, dv.x, -ov.x) : (-2.0f * T) * (dv.x * dv.z);
                    const float y1 = hf ? sg * __builtin_fmaf(dv.w, dv.y, -ov.y) : (-2.0f * T) * (dv.y * dv.z);
                    const uint32_t px0 = split_f16_pair(x0), py0 = split_f16_pair(y0), pe0 = split_f16_pair(e0);
                    const uint32_t px1 = split_f16_pair(x1), py1 = split_f16_pair(y1), pgz = split_f16_pair(gz);
                    // a monomial's slots are (p1, p2 | p1, ...): dwords (X.p1, X.p2), (X.p1, Y.p1), (Y.p1, Y.p2), extra
                    uint4 B0, B1;
                    B0.x = px0; B0.y = (px0 & 0xFFFFu) | (py0 << 16); B0.z = py0; B0.w = hf ? 0x78007800u : pe0;
                    B1.x = px1; B1.y = (px1 & 0xFFFFu) | (py1 << 16); B1.z = py1; B1.w = hf ? (pgz & 0xFFFFu) : pgz;
                    b0[half] = __builtin_bit_cast(h8v, B0);
                    b1[half] = __builtin_bit_cast(h8v, B1);
                }
                const uint4* tabA = reinterpret_cast<const uint4*>(table) + lane;   // tile t, MFMA m: tabA[(2 t + m) * 64]
                // the operands are in registers: the same 2 KB now take the ray records of the exact re-test (a wave's
                // LDS accesses complete in order)
                rayrec[lane]       = make_float4(o.x, o.y, o.z, a);            // two planes of 16-byte records: a wave's stores and its
                rayrec[64u + lane] = make_float4(d.x, d.y, d.z, 0.0f);         // in-order reads are then free of bank conflicts (32-byte records: 2-way)

                RTX_STAMP(8);
                // STREAM: the workgroup's waves copy chunk g + 1 (kStreamGroups x 128 spheres, 8 KB per group, 1 KB per glds16) into one half of
                // the stage while everybody filters chunk g out of the other half; one barrier per chunk orders both the
                // "chunk g has landed" and the "chunk g - 1 is no longer read" directions.
                const uint32_t nwaves     = blockDim.x >> 6;
                const uint32_t wave_u     = uniform(wave);
                const uint32_t stage_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)lds0;
                const uint32_t n_pieces = (P.n_pad_mfma >> 7) * 8u;        // 1 KB pieces of the whole table
                auto stage_chunk = [&](uint32_t g) {                       // chunk g = kStreamGroups groups = 8 kStreamGroups pieces
                    for (uint32_t u = wave_u; u < 8u * kStreamGroups; u += nwaves)
                        if (g * 8u * kStreamGroups + u < n_pieces)
                            glds16(reinterpret_cast<const uint4*>(P.hot_mfma) + ((size_t)g * 8u * kStreamGroups + u) * 64u + lane, stage_addr + (g & 1u) * (kStreamGroups << 13) + (u << 10));
                };
                if (STREAM && P.n_pad_mfma != 0u) stage_chunk(0u);
                h8v A0_next, A1_next;
                if (!STREAM) { A0_next = __builtin_bit_cast(h8v, tabA[0]); A1_next = __builtin_bit_cast(h8v, tabA[64]); }
                for (uint32_t base = 0; base < P.n_pad_mfma; base += 128u) {
                    const uint32_t grp = base >> 7, chunk = grp / kStreamGroups, gic = grp - chunk * kStreamGroups;      // group in chunk
                    const uint4* stage = reinterpret_cast<const uint4*>(lds0) + (((chunk & 1u) * kStreamGroups + gic) << 9) + lane;
                    if (STREAM && gic == 0u) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                        if ((chunk + 1u) * kStreamGroups * 128u < P.n_pad_mfma) stage_chunk(chunk + 1u);
                    }
                    uint32_t word[4];
#pragma unroll
                    for (uint32_t w = 0; w < 2u; ++w) {
                        uint32_t m0 = 0u, m1 = 0u;
#pragma unroll
                        for (uint32_t tt = 0; tt < 2u; ++tt) {
                            h8v A0, A1;
                            if (STREAM) {
                                A0 = __builtin_bit_cast(h8v, stage[(2u * (2u * w + tt)) * 64u]);
                                A1 = __builtin_bit_cast(h8v, stage[(2u * (2u * w + tt) + 1u) * 64u]);
                            } else {
                                A0 = A0_next; A1 = A1_next;
                                const uint32_t nt = (base >> 5) + 2u * w + tt + 1u;           // the table ends with a spare tile
                                A0_next = __builtin_bit_cast(h8v, tabA[(2u * nt) * 64u]);
                                A1_next = __builtin_bit_cast(h8v, tabA[(2u * nt + 1u) * 64u]);
                            }
                            const f16v zero = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
                            f16v h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b0[0], zero, 0, 0, 0);
                            f16v h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, b0[1], zero, 0, 0, 0);
                            h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, b1[0], h0, 0, 0, 0);
                            h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, b1[1], h1, 0, 0, 0);
#pragma unroll
                            for (int j = 0; j < 16; ++j) m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(h0[j]), 31u);
#pragma unroll
                            for (int j = 0; j < 16; ++j) m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(h1[j]), 31u);
                        }
                        // Lanes c and 32 + c trade words (v_permlane32_swap: lanes 32..63 of m0 <-> lanes 0..31 of m1): lane c
                        // gets what lane 32 + c (hf 1) found for ray c, lane 32 + c what lane c (hf 0) found for ray 32 + c.
                        // Afterwards, in every lane, sw[0] is hf 0's word and sw[1] hf 1's word for the lane's own ray.
                        const u2v sw = __builtin_amdgcn_permlane32_swap(m0, m1, false, false);
                        word[2u * w]      = sw[0];
                        word[2u * w + 1u] = sw[1];
                    }
                    RTX_STAMP(2);
                    // A queue entry is (group << 7) | (s << 5) | bit: word s = 2 w + hf' came from lane half hf' and tiles
                    // 2 w, 2 w + 1 of this 128-sphere group, and its bit b (from the LSB) was pushed p = 31 - b th: tile
                    // 2 w + (p >> 4), MFMA row 8 ((p >> 2) & 3) + 4 hf' + (p & 3).  The host lays the table out so that
                    // exactly that row holds sphere (group << 7) | (s << 5) | p, which makes the decode a single xor.
                    const uint32_t grp_code = (base >> 7) << 7;
#pragma unroll
                    for (uint32_t s = 0; s < 4u; ++s) push_word(word[s], grp_code | (s << 5));
                    RTX_STAMP(9);
                }
            }
            RTX_STAMP(2);
            while (wq_count) drain_words(wq_count < 64u ? wq_count : 64u);
            while (q_count) run_batch(q_count < 64u ? q_count : 64u);
            {
                const unsigned long long rr = res[lane];
                closest = __uint_as_float((uint32_t)(rr >> 32));
                best    = (uint32_t)rr;
            }
        }
        RTX_STAMP(3);
        const bool fin = (st == kTrace);        // rays whose _sceneHit is complete: the lanes that (D) and (E) serve (all of `tracing`, except in the grid variant)
        {

            if (VALIDATE) {
                // debug build of the kernel: the reference's full ascending scan next to the filtered result
                float    c2 = 3.402823466e+38f;
                uint32_t b2 = kNoHit;
                for (uint32_t i = 0; i < P.n; ++i) {
                    const float4 s = P.exact[i];
                    const V3    oc = sub(o, mk(s.x, s.y, s.z));
                    const float bq = dot(oc, d);
                    const float cq = dot(oc, oc) - (s.w * s.w);
                    const float disc = bq * bq - a * cq;
                    if (disc > 0) {
                        const float sq = sqrtf(disc);
                        float       t  = (-bq - sq) / a;
                        if (t < c2 && t > 0.001f) { c2 = t; b2 = i; }
                        else {
                            t = (-bq + sq) / a;
                            if (t < c2 && t > 0.001f) { c2 = t; b2 = i; }
                        }
                    }
                }
                const uint64_t bad = __ballot(fin && (b2 != best || (b2 != kNoHit && c2 != closest)));
                c_filter_miss += (uint32_t)__popcll(bad);
            }
        }

        // ---- (D) shade: raytracer.cpp:250-270 + material.cu ------------------------------------------------------
        bool ended = false, absorbed = false, depth_end = false;
        const TraceParams __attribute__((address_space(4)))* const PD = kargs();
        if (fin) {
            if (PD->primary && bounce == 0) PD->primary[(size_t)(s_cur - PD->spp_begin) * PD->npix + pix] = best;
        }
        const bool hit = fin && (best != kNoHit);
        uint32_t   mtype = 0;
        V3         albedo = mk(0, 0, 0), hp = mk(0, 0, 0), hn = mk(0, 0, 0);
        float      blur = 0, ior = 1;
        if (hit) {
            const float4 s  = exact[best];
            const uint4  m0 = PD->mat[2u * best], m1 = PD->mat[2u * best + 1u];
            mtype  = m0.x;
            albedo = mk(__uint_as_float(m0.y), __uint_as_float(m0.z), __uint_as_float(m0.w));
            blur   = __uint_as_float(m1.x);
            ior    = __uint_as_float(m1.y);
            hp     = add(o, scale(closest, d));                          // ray.h:16
            // sphere.cpp:84, normal = (point - centre) / radius: three divisions by the sphere's radius, taken through its double reciprocal
            // from the table (m1.z, m1.w; the identity of `quot`).  That identity leaves out results that round into the subnormal range:
            // a lane with a component below 2^-120 (an exact zero included) divides.
            const V3     pc  = sub(hp, mk(s.x, s.y, s.z));
            const double rdr = __hiloint2double((int)m1.w, (int)m1.z);
            hn = mk((float)((double)pc.x * rdr), (float)((double)pc.y * rdr), (float)((double)pc.z * rdr));
            if (!(__builtin_fabsf(hn.x) >= 7.5231638e-37f && __builtin_fabsf(hn.y) >= 7.5231638e-37f && __builtin_fabsf(hn.z) >= 7.5231638e-37f))
                hn = divs(pc, s.w);
        }

        RTX_STAMP(4);
        // randomInUnitSphere (utils.cu:74-83) for the diffuse and metal lanes -- try t comes from block t of event
        // bounce + 1 -- and the glass lanes' single draw (block 0, word 0), generated COOPERATIVELY: rejection sampling
        // in lockstep would run as many rounds as the unluckiest lane needs tries (5-6 for ~30 lanes at 52 % acceptance)
        // with most lanes idle.  Instead, each round the p pending lanes publish (pixel, sample, slot base, next try) and
        // all 64 lanes work for them: lane i draws try t_j + i / p for pending lane number j = i % p.  An owner takes its
        // lowest accepted try, the same point the sequential loop returns (try 20 is always accepted, utils.cu:80).
        V3    rnd = mk(0, 0, 0);
        float glass_u = 0.0f;
        {
            const bool is_glass = hit && mtype == RTX_MATERIAL_GLASS;
            bool       pending  = hit && (mtype == RTX_MATERIAL_DIFFUSE || mtype == RTX_MATERIAL_METAL || is_glass);
            uint32_t   t = 0;
            uint4* const    own   = reinterpret_cast<uint4*>(wave_lds);               // the ray records are free again
            float4* const   tries = wave_lds + 64;
            uint32_t* const kbest = reinterpret_cast<uint32_t*>(wave_lds + 128);      // and so are the result slots
            for (;;) {
                const uint64_t m = __ballot(pending);
                if (!m) break;
                const uint32_t p    = (uint32_t)__popcll(m);
                const uint32_t rank = rank_in(m);
                if (pending) {
                    own[rank]   = make_uint4(pix, s_cur, (bounce + 1u) * 32u, t | (is_glass ? 0x100u : 0u));
                    kbest[rank] = 0xFFu;
                }
                const float    rp = __builtin_amdgcn_rcpf((float)p);                     // 1 ulp is plenty: see the margins below
                const uint32_t k  = (uint32_t)(((float)lane + 0.5f) * rp);             // lane / p (never within 1/128 of an integer)
                const uint32_t j  = lane - k * p;
                const uint4    ow = own[j];
                const uint32_t tt = (ow.w & 0xFFu) + k;
                const bool     og = (ow.w & 0x100u) != 0u;
                uint32_t r[4];
                philox4x32_10(ow.z + tt, ow.y, ow.x, 0u, P.key0, P.key1, r);
                const float u0 = u32_to_unit(r[0]), u1 = u32_to_unit(r[1]), u2 = u32_to_unit(r[2]);
                const V3    pt = sub(scale(2.0f, mk(u0, u1, u2)), mk(1, 1, 1));
                const bool  accept = og ? (k == 0u) : (tt <= 20u && !(sqlen(pt) >= 1.0f && tt < 20u));
                tries[lane] = make_float4(u0, u1, u2, 0.0f);
                if (accept) __hip_atomic_fetch_min(&kbest[j], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (pending) {
                    const uint32_t kb = kbest[rank];
                    if (kb != 0xFFu) {
                        const float4 w = tries[kb * p + rank];
                        if (is_glass) glass_u = w.x;
                        else          rnd = sub(scale(2.0f, mk(w.x, w.y, w.z)), mk(1, 1, 1));
                        pending = false;
                    } else {
                        t += (uint32_t)((63.5f - (float)rank) * rp) + 1u;            // tries this lane was given: (63 - rank) / p + 1
                    }
                }
            }
        }

        RTX_STAMP(5);
        // |d| and d / |d| once per ray: the metal and glass branches and the background all use them (material.cu:60,
        // :83-88, :120; raytracer.cpp:278), and in lockstep every branch that any lane takes is paid for by all
        const float dlen = sqrt_rn(a);                                      // a = d.d in the order of vector_cuda.h:41
        const V3    dun  = mk(d.x / dlen, d.y / dlen, d.z / dlen);         // vector_cuda.h:45-59
        if (hit) {
            V3   att = albedo, nd_ = d;
            bool ok  = true;
            if (mtype == RTX_MATERIAL_DIFFUSE) {                           // material.cu:47-54
                V3 target = add(add(hp, hn), rnd);
                nd_       = sub(target, hp);
            } else if (mtype == RTX_MATERIAL_METAL) {                      // material.cu:58-65
                V3 vn        = dun;
                V3 reflected = sub(vn, scale(2 * dot(vn, hn), hn));
                nd_          = add(reflected, scale(blur, rnd));
                ok           = dot(nd_, hn) > 0;
            } else if (mtype == RTX_MATERIAL_GLASS) {                      // material.cu:69-105
                V3    reflected = sub(d, scale(2 * dot(d, hn), hn));
                V3    outward;
                float ni, cosine;
                att = mk(1.0f, 1.0f, 1.0f);
                if (dot(d, hn) > 0) {
                    outward = neg(hn);
                    ni      = ior;
                    cosine  = ior * dot(d, hn) / dlen;
                } else {
                    outward = hn;
                    ni      = blur;                                        // 1.0f / ior, divided on the host: a glass sphere's blur word (material.cu:69-105 reads no blur)
                    cosine  = -dot(d, hn) / dlen;
                }
                // _refract, material.cu:118-129
                V3    uv   = dun;
                float dt   = dot(uv, outward);
                float disc = 1.0f - ni * ni * (1.0f - dt * dt);
                V3    refracted = mk(0, 0, 0);
                float prob;
                if (disc > 0) {
                    refracted = sub(scale(ni, sub(uv, scale(dt, outward))), scale(sqrt_rn(disc), outward));
                    // _schlick, material.cu:132-138: pow(float,int) promotes to double
                    const float r0 = albedo.x;                             // ((1 - ior) / (1 + ior))^2 from the host: a glass sphere's first albedo word (its attenuation is (1, 1, 1))
                    double x  = (double)(1.0f - cosine);
                    double x2 = x * x;
                    double x5 = (x2 * x2) * x;
                    prob      = (float)((double)r0 + (double)(1.0f - r0) * x5);
                } else {
                    prob = 1.0f;
                }
                nd_ = (glass_u < prob) ? reflected : refracted;           // material.cu:98-103, the draw made above
            } else {
                ok = false;   // reference: printf + assert(0), material.cu:34-36; uploads reject such tables
            }
            if (recursive && bounce >= P.max_depth) {
                // -r, raytracer.cpp:230-235: `depth < max_depth` is tested before materialScatter; the path is black
                ended = true; depth_end = true;
                thr   = mk(0, 0, 0);
            } else if (ok) {
                if (recursive) {
                    // the outermost attenuation stays in the (otherwise idle) throughput registers, the rest goes to the stack
                    if (bounce == 0u) thr = att;
                    else              att_stack[(size_t)(bounce - 1u) * stack_stride] = make_float4(att.x, att.y, att.z, 0.0f);
                } else {
                    thr = mul(thr, att);                                  // color *= attenuation
                }
                o = hp;
                d = nd_;
                bounce++;
                if (!recursive && bounce >= P.max_depth) { ended = true; depth_end = true; }   // loop bound of raytracer.cpp:250
            } else {
                ended    = true;                                          // break, raytracer.cpp:262-264
                absorbed = true;
                if (recursive) thr = mk(0, 0, 0);                         // raytracer.cpp:233-235
            }
        } else if (fin) {                                                 // _background, raytracer.cpp:276-282
            V3    unit = dun;
            float t    = 0.5f * (unit.y + 1.0f);
            V3    bg   = add(scale(1.0f - t, mk(1.0f, 1.0f, 1.0f)), scale(t, mk(0.5f, 0.7f, 1.0f)));
            if (!recursive) thr = mul(thr, bg);
            ended = true;
            if (recursive) {
                // -r unwinds the recursion: attenuation * (attenuation * (... * background)), innermost product first
                // (raytracer.cpp:232), which rounds differently from the iterative running product
                uint32_t k = bounce > 0u ? bounce - 1u : 0u;                // entries on the stack
                while (k > 0u) {                                            // (divergent loop: lanes leave as they finish)
                    --k;
                    const float4 at = att_stack[(size_t)k * stack_stride];
                    bg = mul(mk(at.x, at.y, at.z), bg);
                }
                thr = bounce > 0u ? mul(thr, bg) : bg;
            }
        }
        // early-terminated-path accounting: one ballot + popcount per cause, kept in wave-uniform registers
        c_miss   += (uint32_t)__popcll(__ballot(fin && !hit));
        c_absorb += (uint32_t)__popcll(__ballot(absorbed));
        c_depth  += (uint32_t)__popcll(__ballot(depth_end));

        RTX_STAMP(6);
        // ---- (E) a finished path is added to its item; a finished item is written out ----------------------------
        if (ended) {
            acc = add(acc, thr);                                          // color += _color(...), raytracer.cpp:184
            s_cur++;
            if (s_cur == s_end) {
#ifdef RTX_WAVE_TIMES
                wt_last_t0 = wt_item_t0; wt_last_rays = item_rays; wt_last_pix = pix; wt_last_len = s_end;
#endif
                kargs()->partials[item] = make_float4(acc.x, acc.y, acc.z, __uint_as_float(item_rays));
                st = kNeedItem;
            } else {
                st = kNeedSample;
            }
        }
        c_samples += (uint32_t)__popcll(__ballot(ended));
        RTX_STAMP(7);
    }

#ifdef RTX_WAVE_TIMES
    if (P.wave_times && lane == 0) {
        unsigned long long* w = P.wave_times + 5u * (blockIdx.x * (blockDim.x >> 6) + wave);
        w[0] = wt_start; w[1] = wt_dry; w[2] = __builtin_amdgcn_s_memrealtime(); w[3] = wt_iters; w[4] = wt_lanes;
    }
    if (P.wave_times) {
        // the lane whose item finished last (largest start + is the last to write): approximated by the lane with the latest finish = any lane
        // whose record was written in the final iterations; take the lane with the most rays among those that finished an item after the wave saw the counter dry
        unsigned long long* w = P.wave_times + 5u * 65536u + 4u * (blockIdx.x * (blockDim.x >> 6) + wave);
        uint32_t best_r = wt_last_rays;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o_ = (uint32_t)__shfl_xor((int)best_r, off); best_r = o_ > best_r ? o_ : best_r; }
        const uint64_t who = __ballot(wt_last_rays == best_r);
        if (lane == (uint32_t)__builtin_ctzll(who)) { w[0] = wt_last_t0; w[1] = wt_last_rays; w[2] = wt_last_pix; w[3] = wt_last_len; }
    }
#endif
#ifdef RTX_WALK_STATS
    if (P.walk_stats) {
        if (lane < 12u && ws_acc)  atomicAdd(&P.walk_stats[lane], (unsigned long long)ws_acc);
        if (lane < 36u && ws_hist) atomicAdd(&P.walk_stats[16u + lane], (unsigned long long)ws_hist);
        if (ws_lin) atomicAdd(&P.walk_stats[64u + lane], (unsigned long long)ws_lin);
    }
#endif
    // wave totals -> global statistics
    if (P.stats) {
        if (FILTER == kFilterGrid && lane_tests) atomicAdd(&P.stats[kStCandidates], (unsigned long long)lane_tests);
        if (lane == 0) {
            atomicAdd(&P.stats[kStSamples], (unsigned long long)c_samples);
            atomicAdd(&P.stats[kStRays], (unsigned long long)c_rays);
            atomicAdd(&P.stats[kStMiss], (unsigned long long)c_miss);
            atomicAdd(&P.stats[kStAbsorb], (unsigned long long)c_absorb);
            atomicAdd(&P.stats[kStDepth], (unsigned long long)c_depth);
            atomicAdd(&P.stats[kStCandidates], (unsigned long long)c_cand);
            atomicAdd(&P.stats[kStFilterMiss], (unsigned long long)c_filter_miss);
#ifdef RTX_PHASE_TIMERS
            for (int k = 0; k < 12; ++k) atomicAdd(&P.stats[kStCount + k], (unsigned long long)ph[k]);
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// accum[p] (+)= sum over chunks, in chunk order, of partials[chunk * npix + p]
// cost (optional): cost[s] receives the largest number of rays (_sceneHit calls) that a pixel of strip s -- pixels 64 s .. 64 s + 63,
// one wave of this kernel -- took in this launch, *cost_total the rays of the whole frame
__global__ void __launch_bounds__(256) reduce_kernel(const float4* __restrict__ partials, float* __restrict__ accum, uint32_t* __restrict__ cost, unsigned long long* __restrict__ cost_total, uint32_t npix, uint32_t nchunks, int accumulate)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    V3 total = mk(0, 0, 0);
    uint32_t rays = 0u;
    if (p < npix)
        for (uint32_t k = 0; k < nchunks; ++k) {
            float4 v = partials[(size_t)k * npix + p];
            total    = add(total, mk(v.x, v.y, v.z));
            rays    += __float_as_uint(v.w);
        }
    if (cost) {
        uint32_t most = rays, sum = rays;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t om = (uint32_t)__shfl_xor((int)most, off), os = (uint32_t)__shfl_xor((int)sum, off);
            most = om > most ? om : most;
            sum += os;
        }
        if (lane_id() == 0u && (p & ~63u) < npix) {
            cost[p >> 6] = most;
            atomicAdd(cost_total, (unsigned long long)sum);
        }
    }
    if (p >= npix) return;
    if (accumulate) total = add(mk(accum[3 * (size_t)p], accum[3 * (size_t)p + 1], accum[3 * (size_t)p + 2]), total);
    accum[3 * (size_t)p + 0] = total.x;
    accum[3 * (size_t)p + 1] = total.y;
    accum[3 * (size_t)p + 2] = total.z;
}

// Diagnostic (rtx_debug_quotient_check, tests/test_gpu_edges.py): the identity behind the grid variant's roots, x / a ==
// (float)((double)x * (1.0 / (double)a)), on pseudo-random operand pairs.  mode 0: arbitrary bit patterns (NaN, infinities, zeros and
// subnormals included); mode 1: the kernel's range, a = d.d in [2^-24, 2^24) and |x| in [2^-40, 2^40); mode 2: a and x with few
// significant bits (quotients close to rounding boundaries).  Results that are NaN on both sides or subnormal / zero on both sides
// count as equal (a root is only ever compared with 0.001 and FLT_MAX).
__global__ void __launch_bounds__(256) quotient_check_kernel(uint32_t seed, uint32_t mode, uint32_t rounds, unsigned long long* __restrict__ bad)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t mism = 0u;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t w[4];
        philox4x32_10(tid, r, seed, mode, 0x9E3779B9u, 0xBB67AE85u, w);
        uint32_t xb = w[0], ab = w[1];
        if (mode == 1u) {
            xb = (w[0] & 0x807FFFFFu) | ((87u + w[2] % 80u) << 23);
            ab = (w[1] & 0x007FFFFFu) | ((103u + w[3] % 48u) << 23);
        } else if (mode == 2u) {
            xb = (w[0] & (0xFFFFFFFFu << (w[2] % 23u))) & 0x807FFFFFu; xb |= (100u + (w[2] >> 8) % 56u) << 23;
            ab = (w[1] & (0xFFFFFFFFu << (w[3] % 23u))) & 0x007FFFFFu; ab |= (100u + (w[3] >> 8) % 56u) << 23;
        }
        if (mode == 3u) {          // sqrt_rn against sqrtf on bit pattern tid * rounds + r: all 2^32 of them for blocks x 256 x rounds = 2^32
            const float v = __uint_as_float(tid * rounds + r);
            const float s1 = sqrtf(v), s2 = sqrt_rn(v);
            mism += (__float_as_uint(s1) == __float_as_uint(s2) || (s1 != s1 && s2 != s2)) ? 0u : 1u;
            continue;
        }
        const float  x = __uint_as_float(xb), a = __uint_as_float(ab);
        const float  q1 = x / a;
        const double rd = 1.0 / (double)a;
        const float  q2 = (float)((double)x * rd);
        const bool   same = __float_as_uint(q1) == __float_as_uint(q2) || (q1 != q1 && q2 != q2)
                         || (__builtin_fabsf(q1) < 1.17549435e-38f && __builtin_fabsf(q2) < 1.17549435e-38f);
        mism += same ? 0u : 1u;
    }
    if (mism) atomicAdd(bad, (unsigned long long)mism);
}

void launch_quotient_check(uint32_t seed, uint32_t mode, uint32_t blocks, uint32_t rounds, unsigned long long* bad, hipStream_t stream)
{
    hipLaunchKernelGGL(quotient_check_kernel, dim3(blocks), dim3(256), 0, stream, seed, mode, rounds, bad);
}

__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ accum, uint32_t n, float value, int accumulate)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) accum[i] = accumulate ? accum[i] + value : value;
}

// raytracer.cpp:187-197 (and :168-171 for the -d borders)
__global__ void __launch_bounds__(256) resolve_kernel(const float* __restrict__ accum, uint32_t* __restrict__ fb, uint32_t cols, uint32_t rows, uint32_t spp_total, uint32_t debug_tile)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cols * rows) return;
    if (debug_tile) {
        uint32_t x = p % cols, y = p / cols, tx = x % debug_tile, ty = y % debug_tile;
        if (ty == 0 || ty == debug_tile - 1 || tx == 0 || tx == debug_tile - 1) { fb[p] = 0xFF000000u; return; }
    }
    const float k  = 1.0f / (float)spp_total;                              // vector_cuda.h:187-195
    const float cr = sqrtf(accum[3 * (size_t)p + 0] * k);
    const float cg = sqrtf(accum[3 * (size_t)p + 1] * k);
    const float cb = sqrtf(accum[3 * (size_t)p + 2] * k);
    const uint32_t r8 = (uint32_t)(uint8_t)(int)(255.99 * (double)cr);
    const uint32_t g8 = (uint32_t)(uint8_t)(int)(255.99 * (double)cg);
    const uint32_t b8 = (uint32_t)(uint8_t)(int)(255.99 * (double)cb);
    fb[p] = (r8 << 24) | (g8 << 16) | (b8 << 8);
}

// ---------------------------------------------------------------------------------------------------------------
template <int FILTER, int TABLE, bool VAL> static const void* kernel_ptr() { return (const void*)trace_kernel<FILTER, TABLE, VAL>; }

static const void* select_kernel(int filter, int table_mode, bool validate)
{
    if (filter == kFilterGrid) return validate ? kernel_ptr<kFilterGrid, kTableGlobal, true>() : kernel_ptr<kFilterGrid, kTableGlobal, false>();
    if (filter == kFilterMfma) {
        if (table_mode == kTableStream) return validate ? kernel_ptr<kFilterMfma, kTableStream, true>() : kernel_ptr<kFilterMfma, kTableStream, false>();
        if (table_mode == kTableLds) return validate ? kernel_ptr<kFilterMfma, kTableLds, true>() : kernel_ptr<kFilterMfma, kTableLds, false>();
        return validate ? kernel_ptr<kFilterMfma, kTableGlobal, true>() : kernel_ptr<kFilterMfma, kTableGlobal, false>();
    }
    if (table_mode == kTableLds) return validate ? kernel_ptr<kFilterValu, kTableLds, true>() : kernel_ptr<kFilterValu, kTableLds, false>();
    return validate ? kernel_ptr<kFilterValu, kTableGlobal, true>() : kernel_ptr<kFilterValu, kTableGlobal, false>();
}

size_t trace_lds_bytes(int filter, int table_mode, uint32_t n, uint32_t n_pad, uint32_t n_pad_mfma, uint32_t block)
{
    const bool   lds_table = table_mode == kTableLds;
    const size_t table = filter == kFilterGrid ? 0 : (lds_table ? (filter == kFilterMfma ? (size_t)(n_pad_mfma + 32) * 64 : (size_t)n_pad * 16) : (table_mode == kTableStream ? kStreamStageBytes : 0));
    const size_t exact = lds_table ? (size_t)((n + 3u) & ~3u) * 16 : 0;
    return kChunkTabBytes + table + exact + (size_t)(block / 64) * wave_lds_bytes(filter);
}

int trace_occupancy(int filter, int table_mode, size_t lds_bytes, uint32_t block)
{
    int blocks = 0;
    if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute(select_kernel(filter, table_mode, false), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, select_kernel(filter, table_mode, false), (int)block, lds_bytes);
    if (e != hipSuccess || blocks < 1) blocks = 1;
    return blocks;
}

hipError_t launch_trace(const TraceParams& p, int filter, int table_mode, bool validate, uint32_t grid, uint32_t block, size_t lds_bytes, hipStream_t stream)
{
    TraceParams copy = p;
    void* args[] = { &copy };
    const void* fn = select_kernel(filter, table_mode, validate);
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    return hipLaunchKernel(fn, dim3(grid), dim3(block), args, lds_bytes, stream);
}

void launch_reduce(const float4* partials, float* accum, uint32_t* cost, unsigned long long* cost_total, uint32_t npix, uint32_t nchunks, bool accumulate, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_kernel, dim3((npix + 255) / 256), dim3(256), 0, stream, partials, accum, cost, cost_total, npix, nchunks, accumulate ? 1 : 0);
}

void launch_fill(float* accum, uint32_t n_floats, float value, bool accumulate, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_kernel, dim3((n_floats + 255) / 256), dim3(256), 0, stream, accum, n_floats, value, accumulate ? 1 : 0);
}

void launch_resolve(const float* accum, uint32_t* fb, uint32_t cols, uint32_t rows, uint32_t spp_total, uint32_t debug_tile, hipStream_t stream)
{
    hipLaunchKernelGGL(resolve_kernel, dim3((cols * rows + 255) / 256), dim3(256), 0, stream, accum, fb, cols, rows, spp_total, debug_tile);
}

} // namespace rtx
