// trace_kernel.hip -- the per-pixel trace loop of avneic/RayTracing as hand-written HIP for gfx950 (CDNA4).
//
// What it replaces (reference file:line):
//   _renderJob's sample loop           raytracer.cpp:157-186      -> persistent waves, one path per lane, lane refill
//   thread_pool tile jobs              raytracer.cpp:93-127       -> work items (sample chunk, pixel) drawn from one
//                                                                   global counter, 256 at a time per wave
//   Camera::getRay                     camera.h:75-86             -> new_sample()
//   _color (iterative)                 raytracer.cpp:243-273      -> the bounce state kept in each lane
//   _sceneHit + sphereHit              raytracer.cpp:285-304,
//                                      sphere.cpp:68-102          -> conservative FMA filter over the LDS-staged SoA
//                                                                   table + exact re-test of the few candidates
//   materialScatter and helpers        material.cu:17-138         -> shade()
//   random / samplers                  utils.cu:65-94             -> Philox4x32-10 counter stream
//
// Numerics: everything that decides a result (camera ray, the exact sphere test, hit point, normal, scatter,
// background, accumulation) is evaluated in the reference's operation order with contraction OFF, IEEE division and
// square root, so a path is bit-identical to the CPU restatement in oracle/rt_oracle.c.  Only the candidate filter
// uses FMAs and a reciprocal square root; it is conservative (never drops a sphere the exact test would accept), so
// it cannot change a result -- only how many exact tests run.
#include "rtx_device.h"

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
__device__ __forceinline__ float len(V3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }         // :41
__device__ __forceinline__ V3    normalized(V3 v)                                                        // :45-59
{
    float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk(v.x / l, v.y / l, v.z / l);
}

// ---------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Random123 constants).  key = seed, counter = (slot, sample, pixel, 0).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
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

enum LaneState : uint32_t { kNeedItem = 0, kNeedSample = 1, kTrace = 2, kDead = 3 };

// ---------------------------------------------------------------------------------------------------------------
template <bool LDS_TABLE>
__global__ void __launch_bounds__(kBlock) trace_kernel(const TraceParams P)
{
    extern __shared__ float4 smem[];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;

    // (1) stage the filter's SoA table into LDS once per workgroup
    const float4* hot;
    uint32_t*     lists;
    if (LDS_TABLE) {
        for (uint32_t i = threadIdx.x; i < P.n_pad; i += kBlock) smem[i] = P.hot[i];
        __syncthreads();
        hot   = smem;
        lists = reinterpret_cast<uint32_t*>(smem + P.n_pad);
    } else {
        hot   = P.hot;
        lists = reinterpret_cast<uint32_t*>(smem);
    }
    uint32_t* my_list = lists + wave * (kListCap * 64u) + lane;   // entry e of this lane: my_list[e * 64]

    // wave-level work reservation: [w_next, w_end) is owned, pf holds the next reservation (lane 0's value)
    uint32_t pf = 0;
    if (lane == 0) pf = atomicAdd(P.work_counter, kWaveBatch);
    uint32_t w_next = uniform(pf);
    uint32_t w_end  = w_next + kWaveBatch;
    if (lane == 0) pf = atomicAdd(P.work_counter, kWaveBatch);

    // per-lane path state
    uint32_t st = kNeedItem, item = 0, pix = 0, s_cur = 0, s_end = 0, bounce = 0;
    V3       o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(1, 1, 1), acc = mk(0, 0, 0);
    uint32_t n_cand = 0;
    // wave-level counters
    uint32_t c_samples = 0, c_miss = 0, c_absorb = 0, c_depth = 0;
    uint64_t c_rays = 0;

    const V3 cam_o = ld3(P.cam.origin), cam_lc = ld3(P.cam.left_corner), cam_h = ld3(P.cam.horizontal), cam_v = ld3(P.cam.vertical);
    const V3 cam_uu = ld3(P.cam.u), cam_vv = ld3(P.cam.v);

    for (;;) {
        // ---- (A) lanes whose item is finished take the next one (persistent tile fetch) -------------------------
        uint64_t need = __ballot(st == kNeedItem);
        if (need) {
            uint32_t cnt = (uint32_t)__popcll(need), r = rank_in(need), avail = w_end - w_next;
            uint32_t it;
            if (cnt <= avail) {
                it = w_next + r;
                w_next += cnt;
            } else {
                uint32_t nb = uniform(pf);
                it     = (r < avail) ? (w_next + r) : (nb + (r - avail));
                w_next = nb + (cnt - avail);
                w_end  = nb + kWaveBatch;
                if (lane == 0) pf = atomicAdd(P.work_counter, kWaveBatch);
            }
            if (st == kNeedItem) {
                if (it < P.n_items) {
                    item           = it;
                    uint32_t chunk = it / P.npix;
                    pix            = it - chunk * P.npix;
                    s_cur          = P.spp_begin + chunk * P.chunk_spp;
                    uint32_t e     = s_cur + P.chunk_spp, lim = P.spp_begin + P.spp_count;
                    s_end          = e < lim ? e : lim;
                    acc            = mk(0, 0, 0);
                    st             = kNeedSample;
                } else {
                    st = kDead;
                }
            }
        }

        // ---- (B) camera rays for lanes starting a sample: raytracer.cpp:177-179 + camera.h:75-86 ------------------
        if (__ballot(st == kNeedSample)) {
            const bool go = (st == kNeedSample);
            uint32_t   px = pix % P.cols, py = pix / P.cols;
            float      b[4];
            rng_block(P, pix, s_cur, 0, 0, b);
            float u = (float)((float)px + b[0]) / (float)P.cols;
            float v = (float)((float)py + b[1]) / (float)P.rows;
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
                float rx = P.cam.lens_radius * dkx, ry = P.cam.lens_radius * dky;
                V3    offset = add(scale(rx, cam_uu), scale(ry, cam_vv));
                o = add(cam_o, offset);
                d = sub(sub(add(add(cam_lc, scale(u, cam_h)), scale(1.0f - v, cam_v)), cam_o), offset);
                thr    = mk(1, 1, 1);
                bounce = 0;
                st     = kTrace;
            }
        }

        const bool     tracing = (st == kTrace);
        const uint64_t tmask   = __ballot(tracing);
        if (!tmask) break;      // every lane is kDead: the queue is drained and all paths have ended
        c_rays += (uint64_t)__popcll(tmask);

        // ---- (C) _sceneHit: conservative filter over all spheres, exact re-test of the candidates ----------------
        const float a = dot(d, d);                                        // sphere.cpp:73
        float       closest = 3.402823466e+38f;                           // FLT_MAX, raytracer.cpp:251
        uint32_t    best = kNoHit;
        {
            const float rs = __builtin_amdgcn_rsqf(a) * (1.0f + kEps);
            const float fx = d.x * rs, fy = d.y * rs, fz = d.z * rs;
            uint32_t    cnt = 0;

            auto process_lists = [&]() {
                for (uint32_t e = 0; __ballot(e < cnt); ++e) {
                    if (e < cnt) {
                        const uint32_t i = my_list[e * 64u];
                        const float4   s = P.exact[i];
                        // sphere.cpp:72-99 in the reference's order
                        const V3    oc = sub(o, mk(s.x, s.y, s.z));
                        const float bq = dot(oc, d);
                        const float cq = dot(oc, oc) - (s.w * s.w);
                        const float disc = bq * bq - a * cq;
                        if (disc > 0) {
                            const float sq = sqrtf(disc);
                            float       t  = (-bq - sq) / a;
                            if (t < closest && t > 0.001f) { closest = t; best = i; }
                            else {
                                t = (-bq + sq) / a;
                                if (t < closest && t > 0.001f) { closest = t; best = i; }
                            }
                        }
                    }
                }
                n_cand += cnt;
                cnt = 0;
            };

            for (uint32_t base = 0; base < P.n_pad; base += 32u) {
                uint32_t m = 0;
#pragma unroll
                for (uint32_t j = 0; j < 32u; ++j) {
                    const float4 s   = hot[base + j];
                    const float  ocx = s.x - o.x, ocy = s.y - o.y, ocz = s.z - o.z;
                    const float  h   = __builtin_fmaf(ocz, fz, __builtin_fmaf(ocy, fy, ocx * fx));
                    const float  q   = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, s.w)));
                    const float  nd  = __builtin_fmaf(-h, h, q);     // < 0  <=>  inflated discriminant > 0
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(nd), 31u);
                }
                if (!tracing) m = 0;
                while (__ballot(m != 0)) {
                    if (m != 0) {
                        const uint32_t j = (uint32_t)__clz((int)m);
                        m &= ~(0x80000000u >> j);
                        my_list[cnt * 64u] = base + j;
                        cnt++;
                    }
                    if (__ballot(cnt == kListCap)) process_lists();
                }
            }
            process_lists();
        }

        // ---- (D) shade: raytracer.cpp:250-270 + material.cu ------------------------------------------------------
        bool ended = false, absorbed = false, depth_end = false;
        if (tracing) {
            if (P.primary && bounce == 0) P.primary[(size_t)(s_cur - P.spp_begin) * P.npix + pix] = best;
        }
        const bool hit = tracing && (best != kNoHit);
        uint32_t   mtype = 0;
        V3         albedo = mk(0, 0, 0), hp = mk(0, 0, 0), hn = mk(0, 0, 0);
        float      blur = 0, ior = 1;
        if (hit) {
            const float4 s  = P.exact[best];
            const uint4  m0 = P.mat[2u * best], m1 = P.mat[2u * best + 1u];
            mtype  = m0.x;
            albedo = mk(__uint_as_float(m0.y), __uint_as_float(m0.z), __uint_as_float(m0.w));
            blur   = __uint_as_float(m1.x);
            ior    = __uint_as_float(m1.y);
            hp     = add(o, scale(closest, d));                          // ray.h:16
            hn     = divs(sub(hp, mk(s.x, s.y, s.z)), s.w);              // sphere.cpp:84
        }

        // randomInUnitSphere (utils.cu:74-83) for the diffuse and metal lanes: try t from block t of event bounce+1
        V3 rnd = mk(0, 0, 0);
        {
            bool     pending = hit && (mtype == RTX_MATERIAL_DIFFUSE || mtype == RTX_MATERIAL_METAL);
            uint32_t t = 0;
            while (__ballot(pending)) {
                float bb[4];
                rng_block(P, pix, s_cur, bounce + 1u, t, bb);
                if (pending) {
                    rnd = sub(scale(2.0f, mk(bb[0], bb[1], bb[2])), mk(1, 1, 1));
                    if (!(sqlen(rnd) >= 1.0f && t < 20)) pending = false;
                }
                t++;
            }
        }

        if (hit) {
            V3   att = albedo, nd_ = d;
            bool ok  = true;
            if (mtype == RTX_MATERIAL_DIFFUSE) {                           // material.cu:47-54
                V3 target = add(add(hp, hn), rnd);
                nd_       = sub(target, hp);
            } else if (mtype == RTX_MATERIAL_METAL) {                      // material.cu:58-65
                V3 vn        = normalized(d);
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
                    cosine  = ior * dot(d, hn) / len(d);
                } else {
                    outward = hn;
                    ni      = 1.0f / ior;
                    cosine  = -dot(d, hn) / len(d);
                }
                // _refract, material.cu:118-129
                V3    uv   = normalized(d);
                float dt   = dot(uv, outward);
                float disc = 1.0f - ni * ni * (1.0f - dt * dt);
                V3    refracted = mk(0, 0, 0);
                float prob;
                if (disc > 0) {
                    refracted = sub(scale(ni, sub(uv, scale(dt, outward))), scale(sqrtf(disc), outward));
                    // _schlick, material.cu:132-138: pow(float,int) promotes to double
                    float  r0 = (1.0f - ior) / (1.0f + ior);
                    r0        = r0 * r0;
                    double x  = (double)(1.0f - cosine);
                    double x2 = x * x;
                    double x5 = (x2 * x2) * x;
                    prob      = (float)((double)r0 + (double)(1.0f - r0) * x5);
                } else {
                    prob = 1.0f;
                }
                float bb[4];
                rng_block(P, pix, s_cur, bounce + 1u, 0, bb);
                nd_ = (bb[0] < prob) ? reflected : refracted;
            } else {
                ok = false;   // reference: printf + assert(0), material.cu:34-36; uploads reject such tables
            }
            if (ok) {
                thr = mul(thr, att);                                      // color *= attenuation
                o   = hp;
                d   = nd_;
                bounce++;
                if (bounce >= P.max_depth) { ended = true; depth_end = true; }   // loop bound of raytracer.cpp:250
            } else {
                ended    = true;                                          // break, raytracer.cpp:262-264
                absorbed = true;
            }
        } else if (tracing) {                                             // _background, raytracer.cpp:276-282
            V3    unit = normalized(d);
            float t    = 0.5f * (unit.y + 1.0f);
            V3    bg   = add(scale(1.0f - t, mk(1.0f, 1.0f, 1.0f)), scale(t, mk(0.5f, 0.7f, 1.0f)));
            thr   = mul(thr, bg);
            ended = true;
        }
        // early-terminated-path accounting: one ballot + popcount per cause, kept in wave-uniform registers
        c_miss   += (uint32_t)__popcll(__ballot(tracing && !hit));
        c_absorb += (uint32_t)__popcll(__ballot(absorbed));
        c_depth  += (uint32_t)__popcll(__ballot(depth_end));

        // ---- (E) a finished path is added to its item; a finished item is written out ----------------------------
        if (ended) {
            acc = add(acc, thr);                                          // color += _color(...), raytracer.cpp:184
            s_cur++;
            if (s_cur == s_end) {
                P.partials[item] = make_float4(acc.x, acc.y, acc.z, 0.0f);
                st = kNeedItem;
            } else {
                st = kNeedSample;
            }
        }
        c_samples += (uint32_t)__popcll(__ballot(ended));
    }

    // wave totals -> global statistics
    if (P.stats) {
        uint32_t v = n_cand;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) {
            atomicAdd(&P.stats[kStSamples], (unsigned long long)c_samples);
            atomicAdd(&P.stats[kStRays], (unsigned long long)c_rays);
            atomicAdd(&P.stats[kStMiss], (unsigned long long)c_miss);
            atomicAdd(&P.stats[kStAbsorb], (unsigned long long)c_absorb);
            atomicAdd(&P.stats[kStDepth], (unsigned long long)c_depth);
            atomicAdd(&P.stats[kStCandidates], (unsigned long long)v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// accum[p] (+)= sum over chunks, in chunk order, of partials[chunk * npix + p]
__global__ void __launch_bounds__(256) reduce_kernel(const float4* __restrict__ partials, float* __restrict__ accum, uint32_t npix, uint32_t nchunks, int accumulate)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    V3 total = mk(0, 0, 0);
    for (uint32_t k = 0; k < nchunks; ++k) {
        float4 v = partials[(size_t)k * npix + p];
        total    = add(total, mk(v.x, v.y, v.z));
    }
    if (accumulate) total = add(mk(accum[3 * (size_t)p], accum[3 * (size_t)p + 1], accum[3 * (size_t)p + 2]), total);
    accum[3 * (size_t)p + 0] = total.x;
    accum[3 * (size_t)p + 1] = total.y;
    accum[3 * (size_t)p + 2] = total.z;
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
size_t trace_lds_bytes(bool lds_table, uint32_t n_pad)
{
    return (lds_table ? (size_t)n_pad * sizeof(float4) : 0) + (size_t)(kBlock / 64) * kListCap * 64 * sizeof(uint32_t);
}

int trace_occupancy(bool lds_table, size_t lds_bytes)
{
    int blocks = 0;
    hipError_t e = lds_table ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trace_kernel<true>, kBlock, lds_bytes)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, trace_kernel<false>, kBlock, lds_bytes);
    if (e != hipSuccess || blocks < 1) blocks = 1;
    return blocks;
}

void launch_trace(const TraceParams& p, bool lds_table, uint32_t grid, size_t lds_bytes, hipStream_t stream)
{
    if (lds_table) hipLaunchKernelGGL(trace_kernel<true>, dim3(grid), dim3(kBlock), lds_bytes, stream, p);
    else           hipLaunchKernelGGL(trace_kernel<false>, dim3(grid), dim3(kBlock), lds_bytes, stream, p);
}

void launch_reduce(const float4* partials, float* accum, uint32_t npix, uint32_t nchunks, bool accumulate, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_kernel, dim3((npix + 255) / 256), dim3(256), 0, stream, partials, accum, npix, nchunks, accumulate ? 1 : 0);
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
