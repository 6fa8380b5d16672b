// rtx_api.hip -- the C ABI of include/rtx.h: context, uploads, launches, read-back.  Host code only; the kernels
// live in trace_kernel.hip.  No CPU fallback exists: every entry point that renders needs a gfx950 device.
#include "rtx_device.h"

#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types only: the library itself is loaded when a group needs it (rccl_load)
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace rtx;

// Scratch of rtx_render_noalloc, the entry point that allocates nothing: work counter (first 256 bytes) + partial sums.
// Every GPU the library runs on holds its own copy from module load until the process ends; in host code the array's name
// is only a handle -- rtx_create asks the runtime for this device's address (hipGetSymbolAddress) and everything else
// uses that pointer.
constexpr size_t kScratchCounterBytes = 256;
constexpr size_t kScratchPartialBytes = 64ull << 20;                         // 4 Mi work items of 16 bytes
static __device__ __align__(256) unsigned char scratch[kScratchCounterBytes + kScratchPartialBytes];

static thread_local std::string g_last_error;
// rtx_render_noalloc: one flag per device, raised while a call of this process uses that device's scratch array
constexpr int kMaxScratchDevices = 64;
static std::atomic<bool> g_scratch_busy[kMaxScratchDevices];
#ifdef RTX_WALK_STATS
static unsigned long long* g_walk_stats = nullptr;      // diagnostic build: 64 counters of the grid walk, one device, one context at a time
#endif
#ifdef RTX_WAVE_TIMES
static unsigned long long* g_wave_times = nullptr;      // diagnostic build: 5 words per wave, one device, one context at a time
#endif

static int fail(int code, const char* fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(RTX_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Tuning knobs (rtx_set_tuning; the RTX_* environment variables give their initial values, read ONCE at rtx_create).
struct Tuning {
    int block = 0;              // forced workgroup size (0: the size that keeps the most waves resident)
    int stream = -1;            // table streaming through the LDS stage: -1 automatic, 0 never, 1 always (MFMA filter)
    int lds_pad = 0;            // extra LDS bytes per workgroup (lowers the occupancy; experiments)
    int items_per_lane = 1024;  // work items per lane that the smallest chunks are sized for (bench frame: 2-sample chunks at the end, 59.2 against 60.8 ms with 256)
    int max_chunks = (int)kMaxPlanChunks;
    int wave_batch = 0;         // forced reservation size (0: automatic)
    int tail_factor = 30;       // how many mean item durations a single item may take (heavy-tailed pixels), see plan_chunks
    int partials_mb = 1024;     // cap of the partial-sum scratch in MiB
    int print_phases = 0;
    int grid_density_pct = 35;  // RTX_FILTER_GRID: spheres per grid cell x 100, applied at the next rtx_upload_scene
    int order = 1;              // cost-ordered work lists for frames the context has traced before (0: never)
    int order_chunks = 16;      // most sample chunks of a cost-ordered launch
    int order_share_pct = 50;   // share of the launch's expected duration that its single longest item may take, in per cent
    int tail_guard = 4;         // reservations are made on demand (not ahead of time) within this many items per resident lane of the list's end
    int walk_park = 10;         // grid variant: the walks of a wave's last walk_park rays are parked in LDS while the wave shades the rest (0: never; book-1 is best at 8, 20 164 spheres at 12)
};

// What identifies "the same frame" for the pixel-cost map: scene, camera and image size.
struct FrameKey {
    uint64_t   scene_version = 0;
    rtx_camera cam{};
    uint32_t   cols = 0, rows = 0;
    bool operator==(const FrameKey& o) const { return scene_version == o.scene_version && cols == o.cols && rows == o.rows && memcmp(&cam, &o.cam, sizeof(cam)) == 0; }
};

// Launch geometry of one (filter, scene) pair: worked out once per scene, not per launch.
struct LaunchPlan {
    bool     valid = false;
    int      table_mode = kTableGlobal;
    uint32_t block = 0;
    size_t   lds_bytes = 0;
    uint32_t max_grid = 0;
};

struct rtx_ctx {
    int         device = -1;
    int         num_cu = 0;
    hipStream_t stream = nullptr;
    hipEvent_t  ev_first = nullptr, ev_last = nullptr;
    hipEvent_t  ev_t0 = nullptr, ev_t1 = nullptr;   // around the most recent trace kernel
    bool        ev_trace = false;
    bool        ev_open = false;      // ev_first recorded since the last sync
    // scene
    uint32_t n = 0, n_pad = 0, n_pad_mfma = 0;
    int      filter = kFilterAuto;  // the linear list through the MFMA filter up to 256 spheres, the uniform grid above (identical results)
    bool     mfma_ok = false;       // the scene fits the f16 range of the MFMA filter
    float    mfma_scale = 1.0f;
    double   mfma_omax = 0.0;       // largest |o| (norm) of a ray origin the MFMA filter's f16 operands can carry
    float4*  hot = nullptr;
    float4*  hot_mfma = nullptr;
    float4*  exact = nullptr;
    uint4*   mat = nullptr;
    GridDesc  grid{};                // RTX_FILTER_GRID: built at upload (rtx_host.cpp, build_grid)
    float4* grid_cells = nullptr; float4* grid_recs = nullptr; uint32_t* grid_big = nullptr;
    bool     have_scene = false, have_camera = false;
    rtx_camera cam{};
    // frame
    uint32_t cols = 0, rows = 0;
    float*   accum = nullptr;  size_t accum_floats = 0;
    bool     accum_borrowed = false;   // caller-owned buffer (rtx_use_accum_buffer)
    float4*  partials = nullptr; size_t partials_items = 0;
    float4*  att_stack = nullptr; size_t att_stack_items = 0;   // RTX_FLAG_RECURSIVE only
    uint32_t* primary = nullptr; size_t primary_items = 0;
    uint32_t* fb = nullptr; size_t fb_items = 0;
    uint32_t* work_counter = nullptr;
    unsigned long long* stats = nullptr;
    Tuning     tune;
    LaunchPlan plan[3];            // by effective filter
    uint32_t   last_tab[kMaxPlanChunks + 1] = { 0 };   // chunk boundaries of the most recent launch (rtx_last_chunk_plan)
    uint32_t   last_nchunks = 0, last_chunk_spp = 0;
    unsigned char* scratch = nullptr;                  // this device's copy of the static scratch array (rtx_render_noalloc)
    // Strip costs and the work order derived from them (see order_for_launch / build_order_list): cost[s] = most rays that a pixel
    // of strip s (pixels 64 s .. 64 s + 63) took in the most recent launch (reduce_kernel writes it); order = the launch's
    // (strip, chunk) batches sorted by expected duration, descending.
    uint64_t  scene_version = 0;
    uint32_t* cost = nullptr;  size_t cost_items = 0;   FrameKey cost_key;  uint32_t cost_spp = 0;   // cost_spp = 0: no map
    bool      cost_fresh = false;                      // cost_total has not been read back yet
    double    cost_mean_rays = 0.0;                    // rays per sample of the frame the cost map describes
    uint32_t* order = nullptr; size_t order_items = 0;  FrameKey order_key; uint32_t order_cost_spp = 0;   // 0: no list
    uint32_t  order_tab[kMaxPlanChunks + 1] = { 0 };   uint32_t order_nchunks = 0;   // the chunk plan the list was built for
    uint32_t* sort_vals = nullptr; size_t sort_vals_items = 0;
    unsigned long long* sort_keys = nullptr; size_t sort_keys_items = 0;     // 2 x batches: keys in, keys out
    void*     sort_tmp = nullptr;  size_t sort_tmp_bytes = 0;
    unsigned long long* cost_total = nullptr;          // device: rays of the whole frame in the most recent launch (reduce_kernel)
    bool      last_ordered = false;                    // the most recent launch ran a cost-ordered work list
    struct Pinned { char* host; size_t bytes; char* dev; };
    std::vector<Pinned> pinned;                        // host memory mapped into the GPU (rtx_host_alloc)
};

extern "C" const char* rtx_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* rtx_version(void) { return "raytracing_amd 0.1 (gfx950)"; }

extern "C" int rtx_create(int device_id, rtx_ctx** out_ctx)
{
    if (!out_ctx) return fail(RTX_ERR_ARG, "rtx_create: out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(RTX_ERR_DEVICE, "rtx_create: no HIP device is visible (this library has no CPU path)");
    if (device_id < 0 || device_id >= count) return fail(RTX_ERR_ARG, "rtx_create: device %d out of range (0..%d)", device_id, count - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(RTX_ERR_DEVICE, "rtx_create: device %d is %s; the kernels are built for gfx950 only", device_id, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device_id));
    rtx_ctx* c = new (std::nothrow) rtx_ctx();
    if (!c) return fail(RTX_ERR_NOMEM, "rtx_create: out of host memory");
    c->device = device_id;
    c->num_cu = prop.multiProcessorCount;
    // the environment is read here and nowhere else; rtx_set_filter / rtx_set_tuning change the values afterwards
    if (const char* f = getenv("RTX_FILTER")) {
        if (strcmp(f, "valu") == 0) c->filter = kFilterValu;
        else if (strcmp(f, "mfma") == 0) c->filter = kFilterMfma;
        else if (strcmp(f, "grid") == 0) c->filter = kFilterGrid;
        else if (strcmp(f, "auto") == 0) c->filter = kFilterAuto;
        else { delete c; return fail(RTX_ERR_ARG, "rtx_create: environment variable RTX_FILTER=%s is refused (valu, mfma, grid or auto)", f); }
    }
    static const struct { const char* env; const char* key; } kEnv[] = {
        { "RTX_BLOCK", "block" }, { "RTX_STREAM", "stream" }, { "RTX_LDS_PAD", "lds_pad" }, { "RTX_ITEMS_PER_LANE", "items_per_lane" },
        { "RTX_MAX_CHUNKS", "max_chunks" }, { "RTX_WAVE_BATCH", "wave_batch" }, { "RTX_TAIL_FACTOR", "tail_factor" },
        { "RTX_PARTIALS_MB", "partials_mb" }, { "RTX_PRINT_PHASES", "print_phases" }, { "RTX_GRID_DENSITY_PCT", "grid_density_pct" },
        { "RTX_TAIL_GUARD", "tail_guard" }, { "RTX_WALK_PARK", "walk_park" }, { "RTX_ORDER", "order" }, { "RTX_ORDER_CHUNKS", "order_chunks" }, { "RTX_ORDER_SHARE_PCT", "order_share_pct" } };
    for (const auto& kv : kEnv)
        if (const char* v = getenv(kv.env)) {
            char* end = nullptr;
            const long val = strtol(v, &end, 10);
            if (end == v || *end != '\0' || val < INT32_MIN || val > INT32_MAX || rtx_set_tuning(c, kv.key, (int)val) != RTX_OK) {
                const std::string why = (end == v || *end != '\0') ? std::string("not an integer") : g_last_error;
                delete c;
                return fail(RTX_ERR_ARG, "rtx_create: environment variable %s=%s is refused (%s)", kv.env, v, why.c_str());
            }
        }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_first);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_last);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_t0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_t1);
    if (e == hipSuccess) e = hipMalloc(&c->work_counter, 256);
    if (e == hipSuccess) e = hipMalloc(&c->stats, (kStCount + 12) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&c->cost_total, sizeof(unsigned long long));
    if (e == hipSuccess) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));      // this device's copy (the device is current)
        c->scratch = static_cast<unsigned char*>(p);
    }
    if (e != hipSuccess) { rtx_destroy(c); return fail(RTX_ERR_DEVICE, "rtx_create: %s", hipGetErrorString(e)); }
    *out_ctx = c;
    return RTX_OK;
}

extern "C" int rtx_destroy(rtx_ctx* c)
{
    if (!c) return RTX_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (const rtx_ctx::Pinned& p : c->pinned) (void)hipHostFree(p.host);
    void* bufs[] = { c->hot, c->hot_mfma, c->exact, c->mat, c->grid_cells, c->grid_recs, c->grid_big, c->accum_borrowed ? nullptr : c->accum, c->partials, c->att_stack, c->primary, c->fb, c->work_counter, c->stats,
                     c->cost, c->order, c->sort_keys, c->sort_vals, c->sort_tmp, c->cost_total };
    for (void* b : bufs) if (b) (void)hipFree(b);
    hipEvent_t evs[] = { c->ev_first, c->ev_last, c->ev_t0, c->ev_t1 };
    for (hipEvent_t ev : evs) if (ev) (void)hipEventDestroy(ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RTX_OK;
}

extern "C" int rtx_upload_scene(rtx_ctx* c, uint32_t n, const float* cx, const float* cy, const float* cz, const float* radius,
                                const uint32_t* mat_type, const float* ar, const float* ag, const float* ab, const float* blur, const float* ior)
{
    if (!c) return fail(RTX_ERR_ARG, "rtx_upload_scene: ctx is NULL");
    if (n > 0 && (!cx || !cy || !cz || !radius || !mat_type || !ar || !ag || !ab || !blur || !ior)) return fail(RTX_ERR_ARG, "rtx_upload_scene: NULL array with n = %u", n);
    if (n > (1u << 24)) return fail(RTX_ERR_ARG, "rtx_upload_scene: n = %u exceeds the supported 2^24 spheres", n);
    for (uint32_t i = 0; i < n; ++i) {
        if (mat_type[i] < RTX_MATERIAL_DIFFUSE || mat_type[i] > RTX_MATERIAL_GLASS)
            return fail(RTX_ERR_ARG, "rtx_upload_scene: sphere %u has unknown material type %u (material.cu:34-36 asserts on it)", i, mat_type[i]);
    }
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n_pad = (n + 31u) & ~31u;
    std::vector<float4> hot(n_pad), exact(std::max<uint32_t>(n, 1u));
    std::vector<uint4>  mat(std::max<uint32_t>(2 * n, 2u));
    for (uint32_t i = 0; i < n_pad; ++i) {
        if (i < n) {
            // the filter sees r^2 inflated by (1 + eps); products in double so that the inflation is not lost to rounding
            const double r2 = (double)radius[i] * (double)radius[i] * (1.0 + (double)kEps);
            hot[i] = make_float4(cx[i], cy[i], cz[i], -(float)std::nextafter((float)r2, INFINITY));
            exact[i] = make_float4(cx[i], cy[i], cz[i], radius[i]);
            uint32_t w[6];
            memcpy(&w[0], &ar[i], 4); memcpy(&w[1], &ag[i], 4); memcpy(&w[2], &ab[i], 4); memcpy(&w[3], &blur[i], 4); memcpy(&w[4], &ior[i], 4);
            if (mat_type[i] == RTX_MATERIAL_GLASS) {
                // what the glass branch would divide for on every hit, divided once here with the same float operations
                // (material.cu:87, :133-134); its albedo and blur are never read (attenuation (1, 1, 1), no fuzz)
                const float inv_ior = 1.0f / ior[i];
                float       r0      = (1.0f - ior[i]) / (1.0f + ior[i]);
                r0                  = r0 * r0;
                memcpy(&w[0], &r0, 4); memcpy(&w[3], &inv_ior, 4);
            }
            const double rdr = 1.0 / (double)radius[i];               // the hit normal's divisions by the radius (sphere.cpp:84), see trace_kernel.hip
            uint32_t rw[2];
            memcpy(rw, &rdr, 8);
            mat[2 * i]     = make_uint4(mat_type[i], w[0], w[1], w[2]);
            mat[2 * i + 1] = make_uint4(w[3], w[4], rw[0], rw[1]);
        } else {
            hot[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0e30f);   // padding: never a candidate
        }
    }
    // MFMA filter (see trace_kernel.hip): per 32-sphere tile two A operands of 64 lanes x 16 B; lane (hf = lane >> 5,
    // r = lane & 31) holds, for sphere 32 t + r, the eight f16 k-slots of slot group (m, hf):
    //   (0,0) CxCx, CyCy, (2^15, 2^15)    (0,1) CzCz, CxCy, (K1, K2)    (1,0) CxCz, CyCz, (Lz1, Lz1)    (1,1) Lx, Ly, (Lz2, 0)
    // with the quadratic monomials Q = scale^2 C_i C_j 2^-15, the linear ones L_i = 2 scale C_i, K = scale^2 K'' 2^-15,
    // K'' = |C|^2 - r^2 - margin; a monomial X next to a monomial Y fills (X1,X1 | X2,Y2 | Y1,Y1).  `scale` is the largest
    // power of two that keeps all of these, and 2 scale (|C| + r) for any bounce origin, inside the f16 range; the camera
    // origin is checked against the same bound per launch (else the VALU filter runs).
    const uint32_t n_pad_mfma = (n + 127u) & ~127u;
    // Range guard: every f16 operand of the filter is bounded by the NORM of a ray origin -- 2 sg C_i and sg (o.d' d'_i - o_i)
    // (the part of o perpendicular to d') by sg |o|, the per-ray constant sg^2 2^-15 (|o|^2' - (o.d')^2) by sg^2 2^-15 |o|^2 -- and
    // a bounce origin lies on a sphere, |o| <= |C| + r.  sg (|C| + 1.01 r) <= 30000 keeps all of them finite in f16
    // (30000^2 2^-15 = 27466); the camera's lens is checked against the same bound per launch.
    double max_on = 0.0, max_k = 0.0;
    bool   finite = true;
    for (uint32_t i = 0; i < n; ++i) {
        const double comp[3] = { std::fabs((double)cx[i]), std::fabs((double)cy[i]), std::fabs((double)cz[i]) };
        const double r = std::fabs((double)radius[i]);
        finite = finite && std::isfinite(comp[0]) && std::isfinite(comp[1]) && std::isfinite(comp[2]) && std::isfinite(r);
        const double C2 = comp[0] * comp[0] + comp[1] * comp[1] + comp[2] * comp[2], r2 = r * r;
        max_on = std::max(max_on, std::sqrt(C2) + 1.01 * r);
        max_k  = std::max(max_k, (1.0 + 3.0 * kGammaSphere) * (C2 + r2));
    }
    int exp2 = 40;
    if (finite && n > 0) {
        while (exp2 > -40) {
            const double sgm = std::ldexp(1.0, exp2);
            if (sgm * max_on <= 30000.0 && sgm * sgm * max_k * std::ldexp(1.0, -15) + 1.0 <= 60000.0) break;
            --exp2;
        }
    } else {
        exp2 = 0;
    }
    const bool   mfma_ok = finite && exp2 > -40;
    const double sgm = std::ldexp(1.0, mfma_ok ? exp2 : 0);
    std::vector<uint4> tabA(((size_t)n_pad_mfma + 32u) * 4u);      // + one spare tile: the kernel prefetches one tile ahead
    auto split2 = [](double x) -> uint32_t {                        // (x1 | x2 << 16), round to nearest
        const _Float16 x1 = (_Float16)x;
        const _Float16 x2 = (_Float16)(x - (double)x1);
        uint16_t a, b; memcpy(&a, &x1, 2); memcpy(&b, &x2, 2);
        return (uint32_t)a | ((uint32_t)b << 16);
    };
    // A-side dwords of a slot group holding monomials X, Y and two extra slots
    auto group = [](uint32_t X, uint32_t Y, uint32_t extra) -> uint4 {
        const uint32_t x1 = X & 0xFFFFu, x2 = X >> 16, y1 = Y & 0xFFFFu, y2 = Y >> 16;
        return make_uint4(x1 | (x1 << 16), x2 | (y2 << 16), y1 | (y1 << 16), extra);
    };
    const double q = sgm * sgm * std::ldexp(1.0, -15);
    const uint4 pad_row = make_uint4(0u, 0u, 0u, split2(60000.0));   // padding: w = 60000 * 2^15, never negative
    for (uint32_t r = 0; r < 32u; ++r) tabA[((size_t)2u * (n_pad_mfma >> 5)) * 64u + 32u + r] = pad_row;   // the spare tile
    for (uint32_t i = 0; i < n_pad_mfma; ++i) {
        uint32_t pxx = 0, pyy = 0, pzz = 0, pxy = 0, pxz = 0, pyz = 0, plx = 0, ply = 0, plz = 0, pk, pone = 0;
        if (i < n && mfma_ok) {
            const double X = cx[i], Y = cy[i], Z = cz[i];
            const double C2 = X * X + Y * Y + Z * Z;
            const double r2 = (double)radius[i] * radius[i];
            const double Kp = C2 - r2;
            const double Kpp = Kp - kGammaSphere * (C2 + r2 + std::fabs(Kp)) - kMfmaAbsMargin / (sgm * sgm);
            pxx = split2(q * X * X); pyy = split2(q * Y * Y); pzz = split2(q * Z * Z);
            pxy = split2(q * X * Y); pxz = split2(q * X * Z); pyz = split2(q * Y * Z);
            plx = split2(2.0 * sgm * X); ply = split2(2.0 * sgm * Y); plz = split2(2.0 * sgm * Z);
            pk  = split2(q * Kpp);
            pone = 0x78007800u;                                      // (2^15, 2^15): carries the per-ray constant
        } else {
            pk = split2(60000.0);                                    // padding: w = 60000 * 2^15, never negative
        }
        // sphere i = (group << 7) | (w << 6) | (hf << 5) | (tt << 4) | (b << 2) | j sits in tile 4 group + 2 w + tt at MFMA row
        // 8 b + 4 hf + j: the (word, bit) position at which the kernel finds its sign then spells i (see trace_kernel.hip)
        const uint32_t t = ((i >> 7) << 2) | (((i >> 6) & 1u) << 1) | ((i >> 4) & 1u);
        const uint32_t r = (((i >> 2) & 3u) << 3) | (((i >> 5) & 1u) << 2) | (i & 3u);
        const uint32_t lz1 = plz & 0xFFFFu, lz2 = plz >> 16;
        tabA[((size_t)2u * t) * 64u + r]            = group(pxx, pyy, pone);
        tabA[((size_t)2u * t) * 64u + 32u + r]      = group(pzz, pxy, pk);
        tabA[((size_t)2u * t + 1u) * 64u + r]       = group(pxz, pyz, lz1 | (lz1 << 16));
        tabA[((size_t)2u * t + 1u) * 64u + 32u + r] = group(plx, ply, lz2);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_scene = false;                                          // a failure below must not leave a half-replaced scene renderable
    if (c->hot_mfma) (void)hipFree(c->hot_mfma);
    c->hot_mfma = nullptr;
    HIP_TRY(hipMalloc(&c->hot_mfma, tabA.size() * sizeof(uint4)));
    HIP_TRY(hipMemcpy(c->hot_mfma, tabA.data(), tabA.size() * sizeof(uint4), hipMemcpyHostToDevice));
    if (c->hot) (void)hipFree(c->hot);
    if (c->exact) (void)hipFree(c->exact);
    if (c->mat) (void)hipFree(c->mat);
    c->hot = nullptr; c->exact = nullptr; c->mat = nullptr;
    HIP_TRY(hipMalloc(&c->hot, std::max<size_t>(n_pad, 32) * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->exact, exact.size() * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->mat, mat.size() * sizeof(uint4)));
    if (n_pad) HIP_TRY(hipMemcpy(c->hot, hot.data(), n_pad * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->exact, exact.data(), exact.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->mat, mat.data(), mat.size() * sizeof(uint4), hipMemcpyHostToDevice));
    // the uniform grid of RTX_FILTER_GRID (a few hundred KB even at 20 k spheres; built for every scene so that the filter can be switched per launch)
    {
        GridHost G;
        build_grid(n, cx, cy, cz, radius, c->tune.grid_density_pct / 100.0, &G);
        void* old[] = { c->grid_cells, c->grid_recs, c->grid_big };
        for (void* b : old) if (b) (void)hipFree(b);
        c->grid_cells = nullptr; c->grid_recs = nullptr; c->grid_big = nullptr;
        // Device form, 32 bytes per walk trip: per cell (first entry, index of the first entry's sphere, entries of tier 0 | up to tier 1
        // << 16, entries up to tier 2) + that sphere's geometry; per entry the geometry of its sphere (a copy) + its index.  The two
        // 16-bit counts bound a cell's inner tiers at 65 535 entries: a table that exceeds that (tens of thousands of spheres through
        // one cell) gets no grid -- every sphere goes on the list that each ray tests in full, the reference's own loop.
        size_t ncells = G.cell_start.empty() ? 0 : G.cell_start.size() - 1;
        for (size_t k = 0; k < ncells; ++k)
            if (G.tier_end[3 * k + 1] - G.cell_start[k] > 0xFFFFu) { ncells = 0; break; }
        if (ncells == 0 && G.desc.dim[0] != 0u) {
            G.desc.dim[0] = G.desc.dim[1] = G.desc.dim[2] = 0u;
            G.big.resize(n);
            for (uint32_t i = 0; i < n; ++i) G.big[i] = i;
            G.desc.n_big = n;
            G.items.clear();
        }
        auto asf = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
        std::vector<float4> cells(2 * ncells), recs(2 * G.items.size());
        for (size_t k = 0; k < ncells; ++k) {
            const uint32_t b = G.cell_start[k], e0 = G.tier_end[3 * k] - b, e1 = G.tier_end[3 * k + 1] - b, e2 = G.tier_end[3 * k + 2] - b;
            cells[2 * k]     = make_float4(asf(b), asf(e2 ? G.items[b] : 0u), asf(e0 | (e1 << 16)), asf(e2));
            cells[2 * k + 1] = e2 ? exact[G.items[b]] : make_float4(0, 0, 0, 0);
        }
        for (size_t k = 0; k < G.items.size(); ++k) { recs[2 * k] = exact[G.items[k]]; recs[2 * k + 1] = make_float4(asf(G.items[k]), 0, 0, 0); }
        HIP_TRY(hipMalloc(&c->grid_cells, std::max<size_t>(cells.size(), 2) * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->grid_recs, std::max<size_t>(recs.size(), 2) * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->grid_big, std::max<size_t>(G.big.size(), 1) * sizeof(uint32_t)));
        if (!cells.empty()) HIP_TRY(hipMemcpy(c->grid_cells, cells.data(), cells.size() * sizeof(float4), hipMemcpyHostToDevice));
        if (!recs.empty()) HIP_TRY(hipMemcpy(c->grid_recs, recs.data(), recs.size() * sizeof(float4), hipMemcpyHostToDevice));
        if (!G.big.empty()) HIP_TRY(hipMemcpy(c->grid_big, G.big.data(), G.big.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->grid = G.desc;
    }
    c->n = n; c->n_pad = n_pad; c->n_pad_mfma = n_pad_mfma; c->have_scene = true;
    c->scene_version++;                                             // pixel costs of the previous scene say nothing about this one
    c->plan[0].valid = c->plan[1].valid = c->plan[2].valid = false;
    c->mfma_ok = mfma_ok; c->mfma_scale = (float)sgm; c->mfma_omax = 30000.0 / sgm;
    return RTX_OK;
}

extern "C" int rtx_set_tuning(rtx_ctx* c, const char* key, int value)
{
    if (!c || !key) return fail(RTX_ERR_ARG, "rtx_set_tuning: NULL argument");
    Tuning& t = c->tune;
    const std::string k = key;
    bool ok = true;
    if (k == "block")               { ok = value == 0 || (value >= 64 && value <= (int)kMaxBlock && value % 64 == 0); if (ok) t.block = value; }
    else if (k == "stream")         { ok = value >= -1 && value <= 1; if (ok) t.stream = value; }
    else if (k == "lds_pad")        { ok = value >= 0 && value <= 160 * 1024; if (ok) t.lds_pad = value; }
    else if (k == "items_per_lane") { ok = value >= 1; if (ok) t.items_per_lane = value; }
    else if (k == "max_chunks")     { ok = value >= 1; if (ok) t.max_chunks = std::min<int>(value, (int)kMaxPlanChunks); }
    else if (k == "wave_batch")     { ok = value == 0 || (value >= 64 && value <= 4096); if (ok) t.wave_batch = value; }
    else if (k == "tail_factor")    { ok = value >= 1 && value <= 1024; if (ok) t.tail_factor = value; }
    else if (k == "partials_mb")    { ok = value >= 1; if (ok) t.partials_mb = value; }
    else if (k == "print_phases")   { t.print_phases = value != 0; }
    else if (k == "grid_density_pct") { ok = value >= 1 && value <= 6400; if (ok) t.grid_density_pct = value; }
    else if (k == "tail_guard")     { ok = value >= 0 && value <= 4096; if (ok) t.tail_guard = value; }
    else if (k == "order")          { ok = value == 0 || value == 1; if (ok) t.order = value; }
    else if (k == "walk_park")      { ok = value >= 0 && value <= 32; if (ok) t.walk_park = value; }
    else if (k == "order_chunks")   { ok = value >= 1 && value <= (int)kMaxPlanChunks; if (ok) t.order_chunks = value; }
    else if (k == "order_share_pct") { ok = value >= 1 && value <= 400; if (ok) t.order_share_pct = value; }
    else return fail(RTX_ERR_ARG, "rtx_set_tuning: unknown knob '%s'", key);
    if (!ok) return fail(RTX_ERR_ARG, "rtx_set_tuning: value %d is out of range for '%s'", value, key);
    c->plan[0].valid = c->plan[1].valid = c->plan[2].valid = false;
    return RTX_OK;
}

extern "C" int rtx_set_filter(rtx_ctx* c, int kind)
{
    if (!c) return fail(RTX_ERR_ARG, "rtx_set_filter: ctx is NULL");
    if (kind != RTX_FILTER_VALU && kind != RTX_FILTER_MFMA && kind != RTX_FILTER_GRID && kind != RTX_FILTER_AUTO) return fail(RTX_ERR_ARG, "rtx_set_filter: unknown filter %d", kind);
    c->filter = kind;
    return RTX_OK;
}

extern "C" int rtx_set_camera(rtx_ctx* c, const rtx_camera* cam)
{
    if (!c || !cam) return fail(RTX_ERR_ARG, "rtx_set_camera: NULL argument");
    c->cam = *cam;
    c->have_camera = true;
    return RTX_OK;
}

template <typename T> static int ensure(T** buf, size_t* have, size_t want)
{
    if (*have >= want && *buf) return RTX_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *have = 0;
    hipError_t e = hipMalloc(buf, want * sizeof(T));
    if (e != hipSuccess) return fail(RTX_ERR_NOMEM, "device allocation of %zu bytes failed: %s", want * sizeof(T), hipGetErrorString(e));
    *have = want;
    return RTX_OK;
}

// Table placement and workgroup size of a (filter, scene) pair: the MFMA filter's table is staged into LDS only while that
// does not cost resident waves -- an L2-fed table with 4 waves/SIMD beats an LDS-resident one with 2 (677 spheres: 17.5
// against 20.9 ms per 100 spp; with equal occupancy LDS wins by 7 %).  The group size is the one that keeps the most waves
// resident (the LDS table is per workgroup, so a larger group amortises it over more waves).  A large table that is not
// LDS-resident is, for the MFMA filter, streamed through a small LDS stage that the waves of a workgroup fill together
// (one L2 read per workgroup instead of one per wave).  The price is a barrier per 128 spheres, i.e. waves in lockstep:
// measured +8 % at 20 164 spheres, +4 % at 4 901, -4 % at 1 601 (profiles/r01h_*), hence the threshold.
static const LaunchPlan& launch_plan(rtx_ctx* c, int filter)
{
    LaunchPlan& lp = c->plan[filter];
    if (lp.valid) return lp;
    const Tuning& t = c->tune;
    const bool lds_fits = filter == kFilterGrid ? false : ((filter == kFilterMfma) ? c->n_pad_mfma <= kMaxLdsSpheresMfma : c->n_pad <= kMaxLdsSpheres);
    auto best_block = [&](int mode, uint32_t* waves_out) -> uint32_t {
        uint32_t best = 0u, best_waves = 0u;
        for (uint32_t cand : { 512u, 256u }) {
            if (t.block && cand != 512u) break;
            const uint32_t blk = t.block ? (uint32_t)t.block : cand;
            const size_t lb = trace_lds_bytes(filter, mode, c->n, c->n_pad, c->n_pad_mfma, blk);
            const uint32_t waves = (uint32_t)trace_occupancy(filter, mode, lb, blk) * (blk / 64u);
            if (waves > best_waves) { best_waves = waves; best = blk; }
        }
        *waves_out = best_waves;
        return best;
    };
    bool stream_ok = filter == kFilterMfma && c->n_pad_mfma >= 3072u;
    if (t.stream >= 0) stream_ok = filter == kFilterMfma && t.stream != 0;
    const int far_mode = stream_ok ? kTableStream : kTableGlobal;
    uint32_t waves_global = 0u, waves_lds = 0u;
    lp.block      = best_block(far_mode, &waves_global);
    lp.table_mode = far_mode;
    if (lds_fits) {
        const uint32_t block_lds = best_block(kTableLds, &waves_lds);
        // the VALU filter reads one sphere per instruction from the table: for it LDS residency matters more than waves
        if (filter == kFilterValu || waves_lds >= waves_global) { lp.table_mode = kTableLds; lp.block = block_lds; }
    }
    lp.lds_bytes = trace_lds_bytes(filter, lp.table_mode, c->n, c->n_pad, c->n_pad_mfma, lp.block) + (size_t)t.lds_pad;
    lp.max_grid  = (uint32_t)c->num_cu * (uint32_t)trace_occupancy(filter, lp.table_mode, lp.lds_bytes, lp.block);
    lp.valid     = true;
    return lp;
}

// Device buffers a launch works in: the context's own (allocated on demand) or the static scratch (rtx_render_noalloc).
struct RenderBuffers {
    float*    accum = nullptr;
    float4*   partials = nullptr; size_t partials_items = 0;
    uint32_t* work_counter = nullptr;
    unsigned long long* stats = nullptr;
    uint32_t* primary = nullptr;
    float4*   att_stack = nullptr; size_t att_stack_items = 0;
    uint32_t* cost = nullptr;               // NULL or npix: rays per pixel of this launch (reduce_kernel)
    unsigned long long* cost_total = nullptr;
};

// Everything of a render that is arithmetic on the host: argument checks, filter choice, launch geometry, chunk plan.
struct RenderSetup {
    TraceParams P{};
    int         filter = kFilterMfma;
    LaunchPlan  lp;
    uint32_t    npix = 0, nchunks = 0, n_items = 0, grid = 0;
    bool        accumulate = false, validate = false, fill_only = false;
    bool        ordered = false;        // the launch runs a cost-ordered batch list (P.order is set once the list exists)
    uint64_t    att_entries = 0;
};

// order_mean_rays > 0: the launch runs the context's cost-ordered work list (c->order), whose pixels took that many rays per
// sample on average when their costs were measured.
static int setup_render(rtx_ctx* c, const char* who, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                        uint64_t seed, uint32_t flags, uint32_t chunk_spp, uint64_t max_items, RenderSetup* out, double order_mean_rays = 0.0)
{
    if (!c) return fail(RTX_ERR_ARG, "%s: ctx is NULL", who);
    if (!c->have_scene || !c->have_camera) return fail(RTX_ERR_STATE, "%s: upload a scene and a camera first", who);
    if (cols == 0 || rows == 0 || spp_count == 0) return fail(RTX_ERR_ARG, "%s: cols, rows and spp_count must be non-zero", who);
    if ((uint64_t)cols * rows > (1ull << 28)) return fail(RTX_ERR_ARG, "%s: image of %u x %u exceeds 2^28 pixels", who, cols, rows);
    if ((uint64_t)spp_begin + spp_count > 0xFFFFFFFFull) return fail(RTX_ERR_ARG, "%s: sample range overflows 32 bits", who);
    if (flags & ~(uint32_t)(RTX_FLAG_ACCUMULATE | RTX_FLAG_RECURSIVE | RTX_FLAG_VALIDATE_FILTER)) return fail(RTX_ERR_ARG, "%s: unknown flag bits 0x%x", who, flags);
    RenderSetup& R = *out;
    R.npix       = cols * rows;
    R.accumulate = (flags & RTX_FLAG_ACCUMULATE) != 0;
    R.validate   = (flags & RTX_FLAG_VALIDATE_FILTER) != 0;
    R.fill_only  = max_depth == 0 && !(flags & RTX_FLAG_RECURSIVE);
    if (R.fill_only) return RTX_OK;

    R.filter = c->filter;
    // RTX_FILTER_AUTO: the linear list through the MFMA filter while it is the faster one, the grid above: up to 256 spheres (two
    // 128-sphere groups) the MFMA filter leads by 3-15 %, from 327 spheres on the grid does (485: 8 732 against 7 275 Msamples/s,
    // 1 601: 8 034 against 3 126; tools/filter_crossover.py, profiles/r02c_grid_filter.txt)
    if (R.filter == kFilterAuto) R.filter = c->n >= kAutoGridSpheres ? kFilterGrid : kFilterMfma;
    if (R.filter == kFilterMfma) {
        // primary rays start on the lens (camera.h:75-86); bounce rays start on spheres, which the scale already covers
        bool fits = c->mfma_ok;
        double reach2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double reach = std::fabs((double)c->cam.origin[k]) + std::fabs((double)c->cam.lens_radius) * (std::fabs((double)c->cam.u[k]) + std::fabs((double)c->cam.v[k]));
            reach2 += reach * reach;
        }
        fits = fits && std::isfinite(reach2) && std::sqrt(reach2) <= c->mfma_omax;       // |o| of any point of the lens
        if (!fits) R.filter = kFilterValu;
    }
    R.lp = launch_plan(c, R.filter);
    const uint64_t lanes = (uint64_t)R.lp.max_grid * R.lp.block;
    const uint64_t npix  = R.npix;

    TraceParams& P = R.P;
    uint32_t first_len;                      // size of the largest chunk: a wave's reservation is served lane by lane, one item after another
    // Cost-ordered launch (order_mean_rays > 0: the frame's strip costs are known): the work list is the launch's (strip, chunk)
    // batches, strip-major over the strips sorted by cost, most expensive first (build_order_list), so the launch ends on the
    // cheapest items of the frame (sky: one ray per sample) and needs far fewer chunks than the geometric plan below, whose
    // many small chunks exist for the tail alone.  What bounds the chunk size is the single most expensive item, which starts
    // at time 0: max_depth rays per sample for the chunk's samples, one ray per wave iteration, against the launch's
    // spp * npix * mean / lanes iterations per lane; it may take order_share_pct of that.  Launches too short for chunks of
    // at least three samples stay unordered: one reservation per 64 items of one or two samples is a trip to the counter
    // every few iterations for every wave (16 spp: 3.4 against 2.9 ms).
    const double order_c_max = order_mean_rays > 0.0 ? std::floor(0.01 * c->tune.order_share_pct * ((double)spp_count * (double)npix * order_mean_rays / (double)lanes) / (double)std::max<uint32_t>(max_depth, 1u)) : 0.0;
    if (chunk_spp == 0 && order_c_max >= 3.0) {
        max_items = std::min<uint64_t>(max_items, (1ull << 31) - 1);
        if (npix > max_items) return fail(RTX_ERR_ARG, "%s: %llu pixels need more than the %llu work items the partial-sum scratch holds", who, (unsigned long long)npix, (unsigned long long)max_items);
        const double c_max = order_c_max;
        const uint64_t allowed = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)c->tune.order_chunks, max_items / npix));
        R.nchunks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>((uint64_t)std::ceil((double)spp_count / c_max), allowed), spp_count));
        for (uint32_t k = 0; k <= R.nchunks; ++k) P.chunk_tab[k] = (uint32_t)((uint64_t)spp_count * k / R.nchunks);
        first_len = P.chunk_tab[1];
        R.ordered = true;         // (enqueue_render builds the list for this plan and sets P.order before the launch)
    } else if (chunk_spp == 0) {
        max_items = std::min<uint64_t>(max_items, (1ull << 31) - 1);
        if (npix > max_items) return fail(RTX_ERR_ARG, "%s: %llu pixels need more than the %llu work items the partial-sum scratch holds", who, (unsigned long long)npix, (unsigned long long)max_items);
        R.nchunks = plan_chunks(spp_count, npix, lanes, (uint32_t)c->tune.items_per_lane, (uint32_t)c->tune.max_chunks, (uint32_t)c->tune.tail_factor, max_items, P.chunk_tab);
        first_len = P.chunk_tab[1];
    } else {
        if (chunk_spp > spp_count) chunk_spp = spp_count;
        R.nchunks = (spp_count + chunk_spp - 1) / chunk_spp;
        first_len = chunk_spp;
    }
    const uint64_t n_items64 = (R.ordered ? (npix + 63) / 64 * 64 : npix) * R.nchunks;     // an ordered list is whole strips: its last one may reach past the image
    if (R.ordered && (npix + 63) / 64 > (1ull << 22)) return fail(RTX_ERR_ARG, "%s: internal: too many strips for an ordered list", who);
    if (n_items64 >= (1ull << 31)) return fail(RTX_ERR_ARG, "%s: %llu work items; raise chunk_spp", who, (unsigned long long)n_items64);
    if (npix * R.nchunks > max_items) return fail(RTX_ERR_ARG, "%s: %llu work items exceed the %llu that the partial-sum scratch holds; raise chunk_spp", who, (unsigned long long)n_items64, (unsigned long long)max_items);
    R.n_items = (uint32_t)n_items64;
    // Reservation size: every trip to the one global counter is a device-scope atomic on one address, and 4096 waves taking
    // 64 one-sample items at a time saturate it (200 spp as 1-sample items: 34.7 ms with 64-item reservations, 24.6 ms with
    // 256); a wave therefore reserves about 512 samples' worth at 512 spheres, at least 64 items (the tail grows with it).
    // Larger tables mean longer bounce iterations and fewer trips, so the target shrinks with the table.
    uint32_t wave_batch = kWaveBatch;
    const uint64_t target = 512ull * 512ull / std::max<uint32_t>(c->n_pad, 512u);
    // (a cost-ordered list keeps 64: a larger reservation would hand several of the most expensive items, which lie next to
    // each other at the head of the list, to the same lanes one after another; its items are never tiny -- at most 16 chunks)
    while (!R.ordered && wave_batch < 512u && (uint64_t)wave_batch * 2u * first_len <= target) wave_batch *= 2u;
    if (c->tune.wave_batch) wave_batch = (uint32_t)c->tune.wave_batch;
    if ((uint64_t)R.n_items + (uint64_t)R.lp.max_grid * (R.lp.block / 64u) * 2u * wave_batch >= (1ull << 32)) return fail(RTX_ERR_ARG, "%s: work counter would overflow; raise chunk_spp", who);

    P.grid = c->grid; P.grid_cells = c->grid_cells; P.grid_recs = c->grid_recs; P.grid_big = c->grid_big;
    P.hot = c->hot; P.hot_mfma = c->hot_mfma; P.n_pad_mfma = c->n_pad_mfma; P.mfma_scale = c->mfma_scale; P.exact = c->exact; P.mat = c->mat; P.n = c->n; P.n_pad = c->n_pad; P.cam = c->cam;
    P.cols = cols; P.rows = rows; P.npix = R.npix; P.inv_cols = 1.0 / (double)(float)cols; P.inv_rows = 1.0 / (double)(float)rows; P.spp_begin = spp_begin; P.spp_count = spp_count; P.chunk_spp = chunk_spp; P.nchunks = R.nchunks;
    P.max_depth = max_depth; P.key0 = (uint32_t)seed; P.key1 = (uint32_t)(seed >> 32); P.flags = flags; P.n_items = R.n_items; P.wave_batch = wave_batch;
    P.tail_guard = (uint32_t)std::min<uint64_t>((uint64_t)c->tune.tail_guard * lanes, 1ull << 30);
    P.walk_park = (uint32_t)c->tune.walk_park;
    if (flags & RTX_FLAG_RECURSIVE) R.att_entries = (uint64_t)std::max<uint32_t>(max_depth, 1u) * R.lp.max_grid * R.lp.block;
    const uint32_t blocks_needed = (uint32_t)((n_items64 + R.lp.block - 1) / R.lp.block);
    R.grid = std::max<uint32_t>(1, std::min<uint32_t>(R.lp.max_grid, blocks_needed));
    memcpy(c->last_tab, P.chunk_tab, sizeof(c->last_tab));
    c->last_nchunks = R.nchunks; c->last_chunk_spp = chunk_spp;
    return RTX_OK;
}

// Puts the launches of one render on the context's stream.  Allocates nothing.
static int launch_render(rtx_ctx* c, RenderSetup& R, const RenderBuffers& B, uint32_t spp_count)
{
    if (!c->ev_open) { HIP_TRY(hipEventRecord(c->ev_first, c->stream)); c->ev_open = true; }
    if (B.stats) HIP_TRY(hipMemsetAsync(B.stats, 0, (kStCount + 12) * sizeof(unsigned long long), c->stream));
    if (R.fill_only) {
        // raytracer.cpp:250: the bounce loop does not run, _color returns (1,1,1) for every sample (-r still traces the primary ray)
        launch_fill(B.accum, R.npix * 3, (float)spp_count, R.accumulate, c->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev_last, c->stream));
        return RTX_OK;
    }
    TraceParams& P = R.P;
    P.partials = B.partials; P.primary = B.primary; P.work_counter = B.work_counter; P.stats = B.stats; P.att_stack = B.att_stack;
#ifdef RTX_WAVE_TIMES
    if (!g_wave_times) { HIP_TRY(hipMalloc(&g_wave_times, 9u * 65536u * 8u)); HIP_TRY(hipMemset(g_wave_times, 0, 9u * 65536u * 8u)); }
    P.wave_times = g_wave_times;
#endif
#ifdef RTX_WALK_STATS
    if (!g_walk_stats) { HIP_TRY(hipMalloc(&g_walk_stats, 128u * 8u)); HIP_TRY(hipMemset(g_walk_stats, 0, 128u * 8u)); }
    P.walk_stats = g_walk_stats;
#endif
    HIP_TRY(hipMemsetAsync(B.work_counter, 0, 256, c->stream));
    HIP_TRY(hipEventRecord(c->ev_t0, c->stream));
    HIP_TRY(launch_trace(P, R.filter, R.lp.table_mode, R.validate, R.grid, R.lp.block, R.lp.lds_bytes, c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_t1, c->stream));
    c->ev_trace = true;
    if (B.cost) HIP_TRY(hipMemsetAsync(B.cost_total, 0, sizeof(unsigned long long), c->stream));
    launch_reduce(B.partials, B.accum, B.cost, B.cost ? B.cost_total : nullptr, R.npix, R.nchunks, R.accumulate, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_last, c->stream));
    return RTX_OK;
}

// The work order of a launch (replaces the role of thread_pool.cpp:344-382's "next free thread takes the next tile": dynamic
// balance, but with the long work handed out first).  A launch of a frame the context has traced before -- same scene, camera
// and size -- knows what every strip of 64 consecutive pixels cost then (the most rays one of its pixels took, reduce_kernel)
// and runs its (strip, chunk) batches in the order of their expected duration, cost x samples of the chunk, longest first:
// the classic longest-processing-time-first rule.  Results do not depend on the order: a pixel's samples are summed per chunk
// and the chunks in chunk order whichever lane runs them.
// order_for_launch: *mean_rays receives the frame's rays per sample when the launch can run ordered, 0 otherwise (first
// launch of a frame, caller-chosen chunks, ordering switched off).
static int order_for_launch(rtx_ctx* c, const FrameKey& key, uint32_t chunk_spp, double* mean_rays)
{
    *mean_rays = 0.0;
    if (chunk_spp != 0) return RTX_OK;
    const size_t npix = (size_t)key.cols * key.rows, nstrips = (npix + 63) / 64;
    const bool have_cost = c->cost && c->cost_spp != 0 && c->cost_key == key && c->cost_items >= nstrips;
    const bool have_order0 = c->order && c->order_cost_spp != 0 && c->order_key == key;
    // (the steady state of a render loop -- a list built from a map at least as good as the new one -- needs no read-back)
    if (have_cost && c->cost_fresh && !(have_order0 && c->cost_mean_rays > 0.0 && c->cost_spp <= c->order_cost_spp)) {
        // the frame's rays per sample, once per cost map.  Waits for the launch that wrote it.
        unsigned long long total = 0;
        HIP_TRY(hipMemcpyAsync(&total, c->cost_total, sizeof(total), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->cost_mean_rays = std::max(1.0, (double)total / ((double)npix * (double)c->cost_spp));
        c->cost_fresh = false;
    }
    const bool have_order = c->order && c->order_cost_spp != 0 && c->order_key == key;
    if (have_cost || have_order) *mean_rays = c->cost_mean_rays;
    return RTX_OK;
}

// key[b] = cost of the batch's strip, val[b] = strip | chunk << 22, for b = strip * nchunks + chunk
static __global__ void __launch_bounds__(256) order_fill_kernel(const uint32_t* __restrict__ cost, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals,
                                                                 uint32_t nstrips, uint32_t nchunks)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nstrips * nchunks) return;
    const uint32_t strip = b / nchunks, chunk = b - strip * nchunks;
    keys[b] = cost[strip];                // the sort is stable: equal keys stay in (strip, chunk) order
    vals[b] = strip | (chunk << 22);
}

// Sorts the launch's batches for the chunk plan in P (on the context's stream, behind the launch that measured the costs).
static int build_order_list(rtx_ctx* c, const FrameKey& key, const TraceParams& P, uint32_t nchunks)
{
    const size_t npix = (size_t)key.cols * key.rows, nstrips = (npix + 63) / 64, nb = nstrips * nchunks;
    const bool have_cost  = c->cost && c->cost_spp != 0 && c->cost_key == key && c->cost_items >= nstrips;
    const bool have_order = c->order && c->order_cost_spp != 0 && c->order_key == key;
    const bool same_plan  = have_order && c->order_nchunks == nchunks && memcmp(c->order_tab, P.chunk_tab, (nchunks + 1) * sizeof(uint32_t)) == 0;
    if (same_plan && !(have_cost && c->cost_spp > c->order_cost_spp)) return RTX_OK;       // the list in hand is the best there is
    if (!have_cost) return 1;                                       // (no costs to sort by: the caller falls back to an unordered launch)
    int rc = ensure(&c->order, &c->order_items, nb);
    if (rc == RTX_OK) rc = ensure(&c->sort_vals, &c->sort_vals_items, nb);
    if (rc == RTX_OK) rc = ensure(&c->sort_keys, &c->sort_keys_items, 2 * nb);
    if (rc) return rc;
    hipLaunchKernelGGL(order_fill_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, c->cost, c->sort_keys, c->sort_vals, (uint32_t)nstrips, nchunks);
    HIP_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, c->sort_keys, c->sort_keys + nb, c->sort_vals, c->order, nb, 0u, 64u, c->stream));
    if (tmp_bytes > c->sort_tmp_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->sort_tmp) (void)hipFree(c->sort_tmp);
        c->sort_tmp = nullptr; c->sort_tmp_bytes = 0;
        if (hipMalloc(&c->sort_tmp, tmp_bytes) != hipSuccess) return fail(RTX_ERR_NOMEM, "device allocation of %zu bytes for the batch sort failed", tmp_bytes);
        c->sort_tmp_bytes = tmp_bytes;
    }
    HIP_TRY(rocprim::radix_sort_pairs_desc(c->sort_tmp, tmp_bytes, c->sort_keys, c->sort_keys + nb, c->sort_vals, c->order, nb, 0u, 64u, c->stream));
    c->order_key = key; c->order_cost_spp = c->cost_spp; c->order_nchunks = nchunks;
    memcpy(c->order_tab, P.chunk_tab, sizeof(c->order_tab));
    return RTX_OK;
}

static int enqueue_render(rtx_ctx* c, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                          uint64_t seed, uint32_t flags, uint32_t chunk_spp, bool want_primary, bool want_stats)
{
    RenderSetup R;
    if (c) HIP_TRY(hipSetDevice(c->device));
    const uint64_t cap_items = c ? ((uint64_t)c->tune.partials_mb << 20) / sizeof(float4) : 0;
    int rc;
    double order_mean_rays = 0.0;
    FrameKey key;
    const bool ordering = c && c->tune.order != 0 && c->have_scene && c->have_camera && cols != 0 && rows != 0 && (uint64_t)cols * rows <= (1ull << 28);
    if (ordering) {
        key.scene_version = c->scene_version; key.cam = c->cam; key.cols = cols; key.rows = rows;
        rc = order_for_launch(c, key, chunk_spp, &order_mean_rays);
        if (rc) return rc;
    }
    rc = setup_render(c, "rtx_render", cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, cap_items, &R, order_mean_rays);
    if (rc) return rc;
    if (R.ordered) {
        rc = build_order_list(c, key, R.P, R.nchunks);
        if (rc < 0) return rc;
        if (rc > 0) {             // no list can be made for this plan: the unordered launch after all
            R = RenderSetup();
            rc = setup_render(c, "rtx_render", cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, cap_items, &R, 0.0);
            if (rc) return rc;
        } else {
            R.P.order = c->order;                                   // (ensure() may have moved it)
        }
    }
    if (R.accumulate && (c->cols != cols || c->rows != rows || !c->accum)) return fail(RTX_ERR_STATE, "rtx_render: RTX_FLAG_ACCUMULATE needs a previous render of the same size");
    if (c->cols != cols || c->rows != rows) { HIP_TRY(hipStreamSynchronize(c->stream)); }
    if (c->accum_borrowed) {
        if (c->accum_floats != (size_t)R.npix * 3) return fail(RTX_ERR_STATE, "rtx_render: the caller's accumulation buffer holds %zu floats, this image needs %zu", c->accum_floats, (size_t)R.npix * 3);
    } else {
        rc = ensure(&c->accum, &c->accum_floats, (size_t)R.npix * 3);
        if (rc) return rc;
    }
    c->cols = cols; c->rows = rows;
    RenderBuffers B;
    B.accum = c->accum; B.work_counter = c->work_counter; B.stats = want_stats ? c->stats : nullptr;
    c->last_ordered = R.P.order != nullptr;
    if (!R.fill_only) {
        if (ordering) {
            // this launch measures the pixels' costs for the next one of the same frame
            rc = ensure(&c->cost, &c->cost_items, ((size_t)R.npix + 63) / 64);
            if (rc) return rc;
            B.cost = c->cost; B.cost_total = c->cost_total;
            c->cost_key = key; c->cost_spp = spp_count; c->cost_fresh = true;
        }
        rc = ensure(&c->partials, &c->partials_items, (size_t)R.n_items);
        if (rc) return rc;
        B.partials = c->partials; B.partials_items = c->partials_items;
        if (want_primary) {
            rc = ensure(&c->primary, &c->primary_items, (size_t)R.npix * spp_count);
            if (rc) return rc;
            B.primary = c->primary;
        }
        if (R.att_entries) {
            // -r multiplies the attenuations innermost first (raytracer.cpp:232), so every lane keeps them until its path ends
            if (R.att_entries * sizeof(float4) > (8ull << 30)) return fail(RTX_ERR_ARG, "rtx_render: RTX_FLAG_RECURSIVE with max_depth %u needs a %llu MiB attenuation stack; the limit is 8 GiB", max_depth, (unsigned long long)(R.att_entries * sizeof(float4) >> 20));
            rc = ensure(&c->att_stack, &c->att_stack_items, (size_t)R.att_entries);
            if (rc) return rc;
            B.att_stack = c->att_stack;
        }
    }
    return launch_render(c, R, B, spp_count);
}

extern "C" int rtx_last_launch_ordered(rtx_ctx* c, int* ordered)
{
    if (!c || !ordered) return fail(RTX_ERR_ARG, "rtx_last_launch_ordered: NULL argument");
    *ordered = c->last_ordered ? 1 : 0;
    return RTX_OK;
}

// Diagnostic (tools/ only, not declared in rtx.h): the strip costs of the most recent launch and the batch list in use.
extern "C" int rtx_debug_order(rtx_ctx* c, uint32_t* cost_host, size_t cost_cap, uint32_t* order_host, size_t order_cap, size_t* n_cost, size_t* n_order)
{
    if (!c) return RTX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t nstrips = ((size_t)c->cost_key.cols * c->cost_key.rows + 63) / 64, nb = nstrips * c->order_nchunks;
    if (n_cost) *n_cost = c->cost ? nstrips : 0;
    if (n_order) *n_order = c->order && c->order_cost_spp ? nb : 0;
    if (c->cost && cost_host && cost_cap >= nstrips) HIP_TRY(hipMemcpy(cost_host, c->cost, nstrips * 4, hipMemcpyDeviceToHost));
    if (c->order && c->order_cost_spp && order_host && order_cap >= nb) HIP_TRY(hipMemcpy(order_host, c->order, nb * 4, hipMemcpyDeviceToHost));
    return RTX_OK;
}

// Diagnostic (tests/ and tools/ only, not declared in rtx.h): counts the operand pairs (blocks x 256 x rounds of them, pseudo-random by
// seed and mode, see quotient_check_kernel) on which the grid variant's root quotient differs from the IEEE division.
extern "C" int rtx_debug_quotient_check(rtx_ctx* c, uint32_t seed, uint32_t mode, uint32_t blocks, uint32_t rounds, unsigned long long* pairs, unsigned long long* mismatches)
{
    if (!c || !pairs || !mismatches || mode > 3u || blocks == 0u || blocks > 65536u || rounds == 0u || rounds > 65536u) return fail(RTX_ERR_ARG, "rtx_debug_quotient_check: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long* bad = nullptr;
    HIP_TRY(hipMalloc(&bad, sizeof(*bad)));
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(*bad), c->stream);
    if (e == hipSuccess) { launch_quotient_check(seed, mode, blocks, rounds, bad, c->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(mismatches, bad, sizeof(*bad), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(bad);
    HIP_TRY(e);
    *pairs = (unsigned long long)blocks * 256ull * rounds;
    return RTX_OK;
}

extern "C" int rtx_last_chunk_plan(rtx_ctx* c, uint32_t* chunk_spp, uint32_t* n_chunks, uint32_t* starts, uint32_t cap)
{
    if (!c || !chunk_spp || !n_chunks) return fail(RTX_ERR_ARG, "rtx_last_chunk_plan: NULL argument");
    *chunk_spp = c->last_chunk_spp;
    *n_chunks  = c->last_nchunks;
    if (c->last_chunk_spp == 0u && c->last_nchunks != 0u) {
        if (!starts || cap < c->last_nchunks + 1u) return fail(RTX_ERR_ARG, "rtx_last_chunk_plan: %u boundaries do not fit into %u entries", c->last_nchunks + 1u, cap);
        memcpy(starts, c->last_tab, (c->last_nchunks + 1u) * sizeof(uint32_t));
    }
    return RTX_OK;
}

extern "C" int rtx_sync(rtx_ctx* c)
{
    if (!c) return fail(RTX_ERR_ARG, "rtx_sync: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTX_OK;
}

extern "C" int rtx_last_kernel_ms(rtx_ctx* c, double* ms)
{
    if (!c || !ms) return fail(RTX_ERR_ARG, "rtx_last_kernel_ms: NULL argument");
    *ms = 0.0;
    if (!c->ev_open) return RTX_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    float f = 0.0f;
    HIP_TRY(hipEventElapsedTime(&f, c->ev_first, c->ev_last));
    *ms = f;
    c->ev_open = false;
    return RTX_OK;
}

extern "C" int rtx_last_trace_ms(rtx_ctx* c, double* ms)
{
    if (!c || !ms) return fail(RTX_ERR_ARG, "rtx_last_trace_ms: NULL argument");
    *ms = 0.0;
    if (!c->ev_trace) return fail(RTX_ERR_STATE, "rtx_last_trace_ms: no trace kernel has been launched");
    HIP_TRY(hipEventSynchronize(c->ev_t1));
    float f = 0.0f;
    HIP_TRY(hipEventElapsedTime(&f, c->ev_t0, c->ev_t1));
    *ms = f;
    return RTX_OK;
}

extern "C" int rtx_use_accum_buffer(rtx_ctx* c, uint32_t cols, uint32_t rows, float* accum_dev)
{
    if (!c) return fail(RTX_ERR_ARG, "rtx_use_accum_buffer: ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->accum_borrowed && c->accum) (void)hipFree(c->accum);
    c->accum = nullptr; c->accum_floats = 0; c->accum_borrowed = false; c->cols = 0; c->rows = 0;
    if (!accum_dev) return RTX_OK;                         // back to a library-owned buffer
    if (cols == 0 || rows == 0) return fail(RTX_ERR_ARG, "rtx_use_accum_buffer: empty image");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, accum_dev) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != c->device)
        return fail(RTX_ERR_ARG, "rtx_use_accum_buffer: pointer is not device memory of GPU %d", c->device);
    c->accum = accum_dev; c->accum_floats = (size_t)cols * rows * 3; c->accum_borrowed = true; c->cols = cols; c->rows = rows;
    return RTX_OK;
}

extern "C" int rtx_render_async(rtx_ctx* c, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                                uint64_t seed, uint32_t flags, uint32_t chunk_spp)
{
    return enqueue_render(c, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, false, false);
}

extern "C" int rtx_render(rtx_ctx* c, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                          uint64_t seed, uint32_t flags, uint32_t chunk_spp, uint32_t* primary_hit_host, rtx_stats* stats)
{
    if (c && c->ev_open) { double dummy; int rc0 = rtx_last_kernel_ms(c, &dummy); if (rc0) return rc0; }
    int rc = enqueue_render(c, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, primary_hit_host != nullptr, stats != nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    double ms = 0.0;
    rc = rtx_last_kernel_ms(c, &ms);
    if (rc) return rc;
#ifdef RTX_WALK_STATS
    if (g_walk_stats && getenv("RTX_PRINT_WALK")) {
        unsigned long long w[128];
        HIP_TRY(hipMemcpy(w, g_walk_stats, sizeof(w), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(g_walk_stats, 0, sizeof(w)));
        auto avg = [&](int a, int b) { return w[b] ? (double)w[a] / (double)w[b] : 0.0; };
        fprintf(stderr, "walk: iterations %llu (%.1f rays each), trips %llu = %.1f per iteration, %.1f rays walking per trip\n", w[0], avg(1, 0), w[2], avg(2, 0), avg(3, 2));
        fprintf(stderr, "  per trip: cell load run %.2f (%.1f lanes)  test %.2f (%.1f lanes)  root %.2f (%.1f lanes)  step %.2f (%.1f lanes)\n",
                avg(4, 2), avg(5, 4), avg(6, 2), avg(7, 6), avg(10, 2), avg(11, 10), avg(8, 2), avg(9, 8));
        fprintf(stderr, "  per ray: trips %.2f  cells %.2f  tests %.2f  roots %.2f   lane-trips / (64 x trips) = %.3f\n", avg(3, 1), avg(5, 1), avg(7, 1), avg(11, 1), (double)w[3] / (64.0 * (double)w[2]));
        const char* names[4] = { "trips of an iteration", "trips of a ray", "cells of a ray", "tests of a ray" };
        for (int h = 0; h < 4; ++h) {
            unsigned long long tot = 0; for (int k = 0; k < 9; ++k) tot += w[16 + 9 * h + k];
            fprintf(stderr, "  %-22s 0: %4.1f %%  1: %4.1f  2-3: %4.1f  4-7: %4.1f  8-15: %4.1f  16-31: %4.1f  32-63: %4.1f  64-127: %4.1f  128+: %4.1f\n", names[h],
                    100.0 * w[16 + 9 * h] / tot, 100.0 * w[17 + 9 * h] / tot, 100.0 * w[18 + 9 * h] / tot, 100.0 * w[19 + 9 * h] / tot, 100.0 * w[20 + 9 * h] / tot,
                    100.0 * w[21 + 9 * h] / tot, 100.0 * w[22 + 9 * h] / tot, 100.0 * w[23 + 9 * h] / tot, 100.0 * w[24 + 9 * h] / tot);
        }
        fprintf(stderr, "  rays by trips 0..63+:");
        for (int k = 0; k < 64; ++k) fprintf(stderr, " %llu", w[64 + k]);
        fprintf(stderr, "\n");
    }
#endif
#ifdef RTX_WAVE_TIMES
    if (g_wave_times && getenv("RTX_PRINT_WAVES")) {
        // diagnostic build: when do the waves run dry and end, relative to the first wave's start (100 MHz clock -> microseconds)
        std::vector<unsigned long long> w(9u * 65536u);
        HIP_TRY(hipMemcpy(w.data(), g_wave_times, w.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0, iters = 0, lanes = 0; size_t nw = 0;
        for (size_t i = 0; i < 65536; ++i) if (w[5 * i + 2]) { t0 = std::min(t0, w[5 * i]); t1 = std::max(t1, w[5 * i + 2]); iters += w[5 * i + 3]; lanes += w[5 * i + 4]; nw++; }
        std::vector<double> start, dry, end;
        for (size_t i = 0; i < 65536; ++i) if (w[5 * i + 2]) { start.push_back((w[5 * i] - t0) * 0.01); dry.push_back(w[5 * i + 1] ? (w[5 * i + 1] - t0) * 0.01 : -1.0); end.push_back((w[5 * i + 2] - t0) * 0.01); }
        auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(p * (v.size() - 1))]; };
        fprintf(stderr, "waves %zu  span %.1f us  iterations/wave %.1f  tracing lanes/iteration %.1f\n", nw, (t1 - t0) * 0.01, (double)iters / nw, (double)lanes / iters);
        fprintf(stderr, "  start us: p0 %.1f p50 %.1f p99 %.1f p100 %.1f\n", pct(start, 0), pct(start, .5), pct(start, .99), pct(start, 1));
        fprintf(stderr, "  dry   us: p0 %.1f p1 %.1f p50 %.1f p99 %.1f p100 %.1f\n", pct(dry, 0), pct(dry, .01), pct(dry, .5), pct(dry, .99), pct(dry, 1));
        fprintf(stderr, "  end   us: p0 %.1f p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f p100 %.1f\n", pct(end, 0), pct(end, .01), pct(end, .1), pct(end, .5), pct(end, .9), pct(end, .99), pct(end, 1));
        // how many waves are still running at the end minus x
        for (double back : { 2000.0, 1000.0, 500.0, 250.0, 100.0, 50.0 }) {
            size_t alive = 0; for (double e : end) if (e > (t1 - t0) * 0.01 - back) alive++;
            fprintf(stderr, "  waves still running %.0f us before the end: %zu\n", back, alive);
        }
        // the 12 waves that ended last: when, and the longest item among the last items of their lanes
        std::vector<std::pair<double, size_t>> byend;
        for (size_t i = 0; i < 65536; ++i) if (w[5 * i + 2]) byend.push_back({ (w[5 * i + 2] - t0) * 0.01, i });
        std::sort(byend.begin(), byend.end());
        for (size_t k = byend.size() > 12 ? byend.size() - 12 : 0; k < byend.size(); ++k) {
            const size_t i = byend[k].second; const unsigned long long* x = &w[5u * 65536u + 4u * i];
            fprintf(stderr, "  wave %5zu end %.1f us dry %.1f iters %llu | longest last item: started %.1f us, %llu rays, pixel %llu (x %llu y %llu), s_end %llu\n", i, byend[k].first, w[5 * i + 1] ? (w[5 * i + 1] - t0) * 0.01 : -1.0, w[5 * i + 3],
                    (double)x[0] * 0.01 + (w[5 * i] - t0) * 0.01, x[1], x[2], x[2] % cols, x[2] / cols, x[3]);
        }
        HIP_TRY(hipMemset(g_wave_times, 0, w.size() * 8));
    }
#endif
    if (primary_hit_host) {
        if (max_depth == 0 && !(flags & RTX_FLAG_RECURSIVE)) memset(primary_hit_host, 0xFF, (size_t)cols * rows * spp_count * sizeof(uint32_t));
        else HIP_TRY(hipMemcpy(primary_hit_host, c->primary, (size_t)cols * rows * spp_count * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    if (stats) {
        unsigned long long h[kStCount + 12];
        HIP_TRY(hipMemcpy(h, c->stats, sizeof(h), hipMemcpyDeviceToHost));
        if (c->tune.print_phases) {
            static const char* names[12] = { "A fetch", "B camera", "C filter tiles", "C exact", "D hit fetch", "D sphere rng", "D material", "E finish", "C filter setup", "C push words", "C peel + enqueue", "C exact batches" };      // (grid variant: C filter tiles = the walk, C exact = full scans, C filter setup = big list, C push words = walk set-up)
            unsigned long long tot = 0;
            for (int k = 0; k < 12; ++k) tot += h[kStCount + k];
            for (int k = 0; k < 12 && tot; ++k) fprintf(stderr, "phase %-18s %6.2f %%\n", names[k], 100.0 * (double)h[kStCount + k] / (double)tot);
        }
        memset(stats, 0, sizeof(*stats));
        if (max_depth == 0 && !(flags & RTX_FLAG_RECURSIVE)) {
            stats->samples = (uint64_t)cols * rows * spp_count;
            stats->term_depth = stats->samples;
        } else {
            stats->samples = h[kStSamples]; stats->rays = h[kStRays]; stats->sphere_tests = h[kStRays] * (uint64_t)c->n;
            stats->term_miss = h[kStMiss]; stats->term_absorb = h[kStAbsorb]; stats->term_depth = h[kStDepth]; stats->candidates = h[kStCandidates];
            stats->filter_misses = h[kStFilterMiss];
        }
        stats->kernel_ms = ms;
    }
    return RTX_OK;
}

extern "C" int rtx_render_noalloc(rtx_ctx* c, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                                  uint64_t seed, uint32_t flags, uint32_t chunk_spp, float* accum_dev)
{
    if (!c || !accum_dev) return fail(RTX_ERR_ARG, "rtx_render_noalloc: NULL argument");
    if (flags & RTX_FLAG_RECURSIVE) return fail(RTX_ERR_ARG, "rtx_render_noalloc: RTX_FLAG_RECURSIVE needs a per-lane attenuation stack, which this entry point has no memory for; use rtx_render");
    if (!c->scratch) return fail(RTX_ERR_STATE, "rtx_render_noalloc: the context has no scratch address");
    if (c->device < 0 || c->device >= kMaxScratchDevices) return fail(RTX_ERR_STATE, "rtx_render_noalloc: device %d is beyond the %d devices the scratch bookkeeping covers", c->device, kMaxScratchDevices);
    HIP_TRY(hipSetDevice(c->device));
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, accum_dev) != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != c->device)
        return fail(RTX_ERR_ARG, "rtx_render_noalloc: accum_dev is not device memory of GPU %d", c->device);
    RenderSetup R;
    int rc = setup_render(c, "rtx_render_noalloc", cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, kScratchPartialBytes / sizeof(float4), &R);
    if (rc) return rc;
    // the allocation behind accum_dev must hold the whole image from accum_dev on (a query, not an allocation)
    if (cols != 0 && rows != 0) {
        hipDeviceptr_t base = nullptr;
        size_t         size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)accum_dev) == hipSuccess) {
            const size_t room = size - (size_t)((const char*)accum_dev - (const char*)base), want = (size_t)cols * rows * 3 * sizeof(float);
            if (room < want) return fail(RTX_ERR_ARG, "rtx_render_noalloc: accum_dev has room for %zu bytes, a %u x %u image needs %zu", room, cols, rows, want);
        } else {
            (void)hipGetLastError();                  // sub-allocators may hand out pointers the runtime cannot size: type and device were checked above
        }
    }
    // one launch at a time per GPU: the scratch array is shared by every context of the process on that device
    bool expected = false;
    if (!g_scratch_busy[c->device].compare_exchange_strong(expected, true, std::memory_order_acquire))
        return fail(RTX_ERR_STATE, "rtx_render_noalloc: another rtx_render_noalloc is running on GPU %d (its calls share one scratch array and must not overlap)", c->device);
    RenderBuffers B;
    B.accum          = accum_dev;
    B.work_counter   = reinterpret_cast<uint32_t*>(c->scratch);
    B.partials       = reinterpret_cast<float4*>(c->scratch + kScratchCounterBytes);
    B.partials_items = kScratchPartialBytes / sizeof(float4);
    rc = launch_render(c, R, B, spp_count);
    const hipError_t e = hipStreamSynchronize(c->stream);      // also after a failed enqueue: nothing of this call may still be running when the flag falls
    g_scratch_busy[c->device].store(false, std::memory_order_release);
    if (rc) return rc;
    if (e != hipSuccess) return fail(RTX_ERR_DEVICE, "rtx_render_noalloc: %s", hipGetErrorString(e));
    c->last_ordered = false;
    return RTX_OK;
}

extern "C" int rtx_accum_device(rtx_ctx* c, float** accum_dev, size_t* n_floats)
{
    if (!c || !accum_dev || !n_floats) return fail(RTX_ERR_ARG, "rtx_accum_device: NULL argument");
    if (!c->accum || c->cols == 0) return fail(RTX_ERR_STATE, "rtx_accum_device: nothing rendered yet");
    *accum_dev = c->accum;
    *n_floats  = (size_t)c->cols * c->rows * 3;
    return RTX_OK;
}

extern "C" int rtx_read_accum(rtx_ctx* c, float* rgb_host, size_t n_floats)
{
    if (!c || !rgb_host) return fail(RTX_ERR_ARG, "rtx_read_accum: NULL argument");
    if (!c->accum || n_floats != (size_t)c->cols * c->rows * 3) return fail(RTX_ERR_ARG, "rtx_read_accum: expected %zu floats", (size_t)c->cols * c->rows * 3);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(rgb_host, c->accum, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return RTX_OK;
}

extern "C" int rtx_write_accum(rtx_ctx* c, uint32_t cols, uint32_t rows, const float* rgb_host, size_t n_floats)
{
    if (!c || !rgb_host) return fail(RTX_ERR_ARG, "rtx_write_accum: NULL argument");
    if (cols == 0 || rows == 0 || n_floats != (size_t)cols * rows * 3) return fail(RTX_ERR_ARG, "rtx_write_accum: size mismatch");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->accum_borrowed) {
        if (c->accum_floats != n_floats) return fail(RTX_ERR_STATE, "rtx_write_accum: size differs from the caller's accumulation buffer");
    } else {
        int rc = ensure(&c->accum, &c->accum_floats, n_floats);
        if (rc) return rc;
    }
    c->cols = cols; c->rows = rows;
    HIP_TRY(hipMemcpy(c->accum, rgb_host, n_floats * sizeof(float), hipMemcpyHostToDevice));
    return RTX_OK;
}

namespace {
struct AccumHeader {          // little endian on disk, 48 bytes, no implicit padding
    char     magic[8];
    uint32_t version, cols, rows, spp_done;
    uint64_t seed;
    uint32_t max_depth, flags, reserved[2];
};
static_assert(sizeof(AccumHeader) == 48, "checkpoint header layout");
}

extern "C" int rtx_save_accum(rtx_ctx* c, const char* path, uint32_t spp_done, uint64_t seed, uint32_t max_depth, uint32_t flags)
{
    if (!c || !path) return fail(RTX_ERR_ARG, "rtx_save_accum: NULL argument");
    if (!c->accum || c->cols == 0) return fail(RTX_ERR_STATE, "rtx_save_accum: nothing rendered yet");
    const size_t n_floats = (size_t)c->cols * c->rows * 3;
    std::vector<float> host(n_floats);
    int rc = rtx_read_accum(c, host.data(), n_floats);
    if (rc) return rc;
    AccumHeader h{};
    memcpy(h.magic, "RTXACCUM", 8);
    h.version = 1; h.cols = c->cols; h.rows = c->rows; h.spp_done = spp_done; h.seed = seed; h.max_depth = max_depth; h.flags = flags;
    // write next to the target, then rename: a crash mid-write must not destroy the previous checkpoint
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(RTX_ERR_ARG, "rtx_save_accum: cannot open %s", tmp.c_str());
    const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(host.data(), sizeof(float), n_floats, f) == n_floats;
    if (fclose(f) != 0 || !ok) { remove(tmp.c_str()); return fail(RTX_ERR_ARG, "rtx_save_accum: short write to %s", tmp.c_str()); }
    if (rename(tmp.c_str(), path) != 0) return fail(RTX_ERR_ARG, "rtx_save_accum: cannot rename %s to %s", tmp.c_str(), path);
    return RTX_OK;
}

extern "C" int rtx_load_accum(rtx_ctx* c, const char* path, uint32_t* cols, uint32_t* rows, uint32_t* spp_done, uint64_t* seed, uint32_t* max_depth, uint32_t* flags)
{
    if (!c || !path) return fail(RTX_ERR_ARG, "rtx_load_accum: NULL argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(RTX_ERR_ARG, "rtx_load_accum: cannot open %s", path);
    AccumHeader h{};
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "RTXACCUM", 8) != 0 || h.version != 1) { fclose(f); return fail(RTX_ERR_ARG, "rtx_load_accum: %s is not a version-1 accumulation checkpoint", path); }
    if (h.cols == 0 || h.rows == 0 || (uint64_t)h.cols * h.rows > (1ull << 28)) { fclose(f); return fail(RTX_ERR_ARG, "rtx_load_accum: %s holds an image of %u x %u", path, h.cols, h.rows); }
    const size_t n_floats = (size_t)h.cols * h.rows * 3;
    std::vector<float> host(n_floats);
    const size_t got = fread(host.data(), sizeof(float), n_floats, f);
    fclose(f);
    if (got != n_floats) return fail(RTX_ERR_ARG, "rtx_load_accum: %s is truncated (%zu of %zu floats)", path, got, n_floats);
    int rc = rtx_write_accum(c, h.cols, h.rows, host.data(), n_floats);
    if (rc) return rc;
    if (cols) *cols = h.cols;
    if (rows) *rows = h.rows;
    if (spp_done) *spp_done = h.spp_done;
    if (seed) *seed = h.seed;
    if (max_depth) *max_depth = h.max_depth;
    if (flags) *flags = h.flags;
    return RTX_OK;
}

extern "C" int rtx_resolve_rgbx8(rtx_ctx* c, uint32_t spp_total, uint32_t debug_tile, uint32_t* framebuffer_host)
{
    if (!c || !framebuffer_host) return fail(RTX_ERR_ARG, "rtx_resolve_rgbx8: NULL argument");
    if (!c->accum || c->cols == 0) return fail(RTX_ERR_STATE, "rtx_resolve_rgbx8: nothing rendered yet");
    if (spp_total == 0) return fail(RTX_ERR_ARG, "rtx_resolve_rgbx8: spp_total is 0");
    HIP_TRY(hipSetDevice(c->device));
    const size_t npix = (size_t)c->cols * c->rows;
    // a framebuffer inside memory from rtx_host_alloc is written by the kernel itself, over the host link
    for (const rtx_ctx::Pinned& p : c->pinned) {
        const char* fbh = reinterpret_cast<const char*>(framebuffer_host);
        if (fbh >= p.host && fbh + npix * sizeof(uint32_t) <= p.host + p.bytes) {
            launch_resolve(c->accum, reinterpret_cast<uint32_t*>(p.dev + (fbh - p.host)), c->cols, c->rows, spp_total, debug_tile, c->stream);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(c->stream));
            return RTX_OK;
        }
    }
    int rc = ensure(&c->fb, &c->fb_items, npix);
    if (rc) return rc;
    launch_resolve(c->accum, c->fb, c->cols, c->rows, spp_total, debug_tile, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(framebuffer_host, c->fb, npix * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RTX_OK;
}

extern "C" int rtx_host_alloc(rtx_ctx* c, size_t bytes, void** host)
{
    if (!c || !host || bytes == 0) return fail(RTX_ERR_ARG, "rtx_host_alloc: NULL argument or empty size");
    *host = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess || !p) { (void)hipGetLastError(); return fail(RTX_ERR_NOMEM, "rtx_host_alloc: hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e)); }
    void* dev = nullptr;
    e = hipHostGetDevicePointer(&dev, p, 0);
    if (e != hipSuccess || !dev) { (void)hipHostFree(p); return fail(RTX_ERR_DEVICE, "rtx_host_alloc: hipHostGetDevicePointer failed: %s", hipGetErrorString(e)); }
    c->pinned.push_back({ static_cast<char*>(p), bytes, static_cast<char*>(dev) });
    *host = p;
    return RTX_OK;
}

extern "C" int rtx_host_free(rtx_ctx* c, void* host)
{
    if (!c || !host) return fail(RTX_ERR_ARG, "rtx_host_free: NULL argument");
    for (size_t i = 0; i < c->pinned.size(); ++i)
        if (c->pinned[i].host == host) {
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipHostFree(host));
            c->pinned.erase(c->pinned.begin() + (long)i);
            return RTX_OK;
        }
    return fail(RTX_ERR_ARG, "rtx_host_free: %p was not allocated through this context", host);
}

// ---------------------------------------------------------------------------------------------------------------
// Several GPUs of one node behind the C ABI: one process, one member context per device, the
// samples of every pixel split over the members by global sample index, ONE RCCL all-reduce (sum, float32) of the
// accumulation buffers over xGMI, resolve on member 0.  The reference has no counterpart (it is single-device).
// One host thread per member beyond the first: the members' launches are queued side by side instead of one device after
// the other (a launch sequence is about ten runtime calls, 40-60 us: queued from one thread, the eighth GPU of a node would
// start 0.4 ms after the first -- 5 % of a 62-sample share).  A worker runs one task at a time and keeps the error message
// of a failed task (rtx_last_error is per thread) for the caller.
struct GroupWorker {
    std::thread             th;
    std::mutex              m;
    std::condition_variable cv;
    std::function<int()>    task;
    bool                    has_task = false, done = true, quit = false;
    int                     rc = 0;
    std::string             err;
    void loop()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_task || quit; });
            if (quit) return;
            std::function<int()> t = std::move(task);
            has_task = false;
            lk.unlock();
            const int r = t();
            const std::string e = r ? g_last_error : std::string();
            lk.lock();
            rc = r; err = e; done = true;
            cv.notify_all();
        }
    }
    void start() { th = std::thread([this] { loop(); }); }
    void post(std::function<int()> t)
    {
        std::lock_guard<std::mutex> lk(m);
        task = std::move(t); has_task = true; done = false;
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); }
        if (th.joinable()) th.join();
    }
};

struct rtx_group {
    std::vector<rtx_ctx*>   members;
    std::vector<std::unique_ptr<GroupWorker>> workers;   // workers[r - 1] serves member r
    double                  last_enqueue_us = 0.0;       // host time of the most recent group_enqueue_render (diagnostic)
    // fn(r) for every member, member 0 on the calling thread, the others on their workers; the first failure's code and message
    int for_each_member(const std::function<int(uint32_t)>& fn)
    {
        const uint32_t world = (uint32_t)members.size();
        for (uint32_t r = 1; r < world; ++r) workers[r - 1]->post([&fn, r] { return fn(r); });
        int rc = fn(0);
        std::string err = rc ? g_last_error : std::string();
        for (uint32_t r = 1; r < world; ++r) {
            const int rr = workers[r - 1]->wait();
            if (rr && !rc) { rc = rr; err = workers[r - 1]->err; }
        }
        if (rc) g_last_error = err;
        return rc;
    }
    std::vector<ncclComm_t> comms;        // empty for a group of one: its all-reduce is the identity
    bool                    loopback = false;   // RTX_GROUP_LOOPBACK=1: several members on ONE device, exchange without RCCL (rehearsal)
    hipEvent_t              ex0 = nullptr, ex1 = nullptr;   // on member 0's stream, around its part of the most recent exchange
    bool                    ex_timed = false;
};

// the exchange of a loopback group: dst += src (same device)
static __global__ void __launch_bounds__(256) accum_add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = dst[i] + src[i];
}

// RCCL is resolved when the first group that needs it is created (dlopen + dlsym), so that a process that renders on one GPU -- the
// common case -- neither loads librccl.so nor pays its start-up.  The seven entry points the exchange uses:
struct RcclApi {
    void* lib = nullptr;
    const char*  (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mutex;

static int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return RTX_OK;
    const char* names[] = { getenv("RTX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
    void* lib = nullptr;
    for (const char* nm : names) if (nm && *nm && (lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return fail(RTX_ERR_DEVICE, "a group of several GPUs needs RCCL: librccl.so could not be loaded (%s); set RTX_RCCL_LIB to its path", dlerror());
    RcclApi a;
    a.lib = lib;
    bool ok = true;
    auto sym = [&](const char* nm) { void* p = dlsym(lib, nm); if (!p) ok = false; return p; };
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.CommInitAll    = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
    a.CommDestroy    = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.GroupStart     = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd       = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.AllReduce      = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.Reduce         = reinterpret_cast<decltype(a.Reduce)>(sym("ncclReduce"));
    if (!ok) { dlclose(lib); return fail(RTX_ERR_DEVICE, "librccl.so lacks one of the entry points the exchange uses"); }
    g_rccl = a;
    return RTX_OK;
}

#define NCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(RTX_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// Contiguous share of [spp_begin, spp_begin + spp_count) for member `rank` of `world`; remainder samples go to the lowest
// ranks (the rule of raytracing_amd/multi_gpu.py spp_range, which the one-process-per-GPU host uses).
extern "C" void rtx_spp_range(uint32_t rank, uint32_t world, uint32_t spp_begin, uint32_t spp_count, uint32_t* begin, uint32_t* count)
{
    if (world == 0) world = 1;
    const uint32_t base = spp_count / world, rem = spp_count % world;
    if (count) *count = base + (rank < rem ? 1u : 0u);
    if (begin) *begin = spp_begin + rank * base + (rank < rem ? rank : rem);
}

extern "C" int rtx_group_destroy(rtx_group* g)
{
    if (!g) return RTX_OK;
    for (auto& w : g->workers) w->stop();
    for (size_t i = 0; i < g->comms.size(); ++i) {
        if (g->members[i]) (void)hipSetDevice(g->members[i]->device);
        if (g->comms[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g->comms[i]);
    }
    if (!g->members.empty() && g->members[0]) (void)hipSetDevice(g->members[0]->device);
    if (g->ex0) (void)hipEventDestroy(g->ex0);
    if (g->ex1) (void)hipEventDestroy(g->ex1);
    for (rtx_ctx* c : g->members) (void)rtx_destroy(c);
    delete g;
    return RTX_OK;
}

extern "C" int rtx_group_create(const int* device_ids, int n_devices, rtx_group** out)
{
    if (!out) return fail(RTX_ERR_ARG, "rtx_group_create: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return fail(RTX_ERR_ARG, "rtx_group_create: need 1..64 device ids");
    // RTX_GROUP_LOOPBACK=1: a rehearsal of groups of several members on a box with one GPU -- every id must name the same
    // device, the members render their shares one after another on it and the exchange is a plain sum into member 0 (no
    // RCCL: it refuses two ranks on one device).  Everything else -- the split, the invariant, progressive passes -- is the
    // code that a multi-GPU group runs.
    const char* lb = getenv("RTX_GROUP_LOOPBACK");
    const bool  loopback = lb && atoi(lb) != 0 && n_devices > 1;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) {
            if (!loopback && device_ids[i] == device_ids[j]) return fail(RTX_ERR_ARG, "rtx_group_create: device %d is listed twice", device_ids[i]);
            if (loopback && device_ids[i] != device_ids[j]) return fail(RTX_ERR_ARG, "rtx_group_create: RTX_GROUP_LOOPBACK needs every member on the same device");
        }
    rtx_group* g = new (std::nothrow) rtx_group();
    if (!g) return fail(RTX_ERR_NOMEM, "rtx_group_create: out of host memory");
    g->loopback = loopback;
    g->members.assign((size_t)n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) {
        const int rc = rtx_create(device_ids[i], &g->members[(size_t)i]);
        if (rc) { rtx_group_destroy(g); return rc; }           // rtx_create has set the message
    }
    for (int i = 1; i < n_devices; ++i) {
        g->workers.emplace_back(new GroupWorker());
        g->workers.back()->start();
    }
    // RTX_GROUP_RCCL=1 makes a group of ONE device go through RCCL too (a communicator of one rank): the rehearsal of the
    // exchange on a single-GPU box; without it a group of one never touches RCCL
    const char* force = getenv("RTX_GROUP_RCCL");
    if (!loopback && (n_devices > 1 || (force && atoi(force) != 0))) {
        const int lrc = rccl_load();
        if (lrc) { rtx_group_destroy(g); return lrc; }
        g->comms.assign((size_t)n_devices, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n_devices, device_ids);
        if (r != ncclSuccess) {
            g->comms.clear();
            rtx_group_destroy(g);
            return fail(RTX_ERR_DEVICE, "rtx_group_create: ncclCommInitAll over %d devices failed: %s", n_devices, g_rccl.GetErrorString(r));
        }
    }
    if (hipSetDevice(g->members[0]->device) != hipSuccess || hipEventCreate(&g->ex0) != hipSuccess || hipEventCreate(&g->ex1) != hipSuccess) {
        rtx_group_destroy(g);
        return fail(RTX_ERR_DEVICE, "rtx_group_create: cannot create the exchange's events");
    }
    *out = g;
    return RTX_OK;
}

extern "C" int rtx_group_size(const rtx_group* g) { return g ? (int)g->members.size() : 0; }

extern "C" rtx_ctx* rtx_group_member(rtx_group* g, int rank)
{
    if (!g || rank < 0 || rank >= (int)g->members.size()) { (void)fail(RTX_ERR_ARG, "rtx_group_member: rank out of range"); return nullptr; }
    return g->members[(size_t)rank];
}

extern "C" int rtx_group_upload_scene(rtx_group* g, uint32_t n, const float* cx, const float* cy, const float* cz, const float* radius,
                                      const uint32_t* mat_type, const float* ar, const float* ag, const float* ab, const float* blur, const float* ior)
{
    if (!g) return fail(RTX_ERR_ARG, "rtx_group_upload_scene: group is NULL");
    // (every member builds its own tables and grid: side by side on the members' host threads)
    return g->for_each_member([&](uint32_t r) { return rtx_upload_scene(g->members[r], n, cx, cy, cz, radius, mat_type, ar, ag, ab, blur, ior); });
}

extern "C" int rtx_group_set_camera(rtx_group* g, const rtx_camera* cam)
{
    if (!g) return fail(RTX_ERR_ARG, "rtx_group_set_camera: group is NULL");
    for (rtx_ctx* c : g->members) {
        const int rc = rtx_set_camera(c, cam);
        if (rc) return rc;
    }
    return RTX_OK;
}

static int member_enqueue_render(rtx_group* g, uint32_t r, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                                 uint64_t seed, uint32_t flags, uint32_t chunk_spp, bool want_stats)
{
    const uint32_t world = (uint32_t)g->members.size();
    rtx_ctx* c = g->members[r];
    uint32_t begin = 0, count = 0;
    rtx_spp_range(r, world, spp_begin, spp_count, &begin, &count);
    HIP_TRY(hipSetDevice(c->device));                        // every per-member call below runs with the member's device current
    if (c->ev_open) { double dummy; const int rc0 = rtx_last_kernel_ms(c, &dummy); if (rc0) return rc0; }
    if (count) {
        const int rc = enqueue_render(c, cols, rows, begin, count, max_depth, seed, flags, chunk_spp, false, want_stats);
        if (rc) return rc;
    } else if (!(flags & RTX_FLAG_ACCUMULATE)) {
        // more devices than samples: this member contributes zeros (it still takes part in the all-reduce)
        if (c->cols != cols || c->rows != rows) HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->accum_borrowed) {
            if (c->accum_floats != (size_t)cols * rows * 3) return fail(RTX_ERR_STATE, "rtx_group_render: a member's borrowed accumulation buffer has the wrong size");
        } else {
            const int rc = ensure(&c->accum, &c->accum_floats, (size_t)cols * rows * 3);
            if (rc) return rc;
        }
        c->cols = cols; c->rows = rows;
        HIP_TRY(hipMemsetAsync(c->accum, 0, (size_t)cols * rows * 3 * sizeof(float), c->stream));
        if (want_stats) HIP_TRY(hipMemsetAsync(c->stats, 0, (kStCount + 12) * sizeof(unsigned long long), c->stream));
    } else if (c->cols != cols || c->rows != rows || !c->accum) {
        return fail(RTX_ERR_STATE, "rtx_group_render: RTX_FLAG_ACCUMULATE needs a previous render of the same size");
    }
    return RTX_OK;
}

// Queues every member's share of the sample range on its own stream -- every member from its own host thread -- and waits
// for nothing on the devices: the launches of all GPUs run side by side.
static int group_enqueue_render(rtx_group* g, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                                uint64_t seed, uint32_t flags, uint32_t chunk_spp, bool want_stats)
{
    if (cols == 0 || rows == 0 || spp_count == 0) return fail(RTX_ERR_ARG, "rtx_group_render: cols, rows and spp_count must be non-zero");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = RTX_OK;
    if (getenv("RTX_GROUP_SERIAL")) {        // diagnostic: one device after the other from the calling thread (round 2's behaviour)
        for (uint32_t r = 0; r < (uint32_t)g->members.size() && rc == RTX_OK; ++r) rc = member_enqueue_render(g, r, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, want_stats);
    } else {
        rc = g->for_each_member([&](uint32_t r) { return member_enqueue_render(g, r, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, want_stats); });
    }
    g->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    (void)hipSetDevice(g->members[0]->device);
    return rc;
}

extern "C" int rtx_group_render(rtx_group* g, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                                uint64_t seed, uint32_t flags, uint32_t chunk_spp, rtx_stats* stats)
{
    if (!g) return fail(RTX_ERR_ARG, "rtx_group_render: group is NULL");
    const uint32_t world = (uint32_t)g->members.size();
    const int rc_q = group_enqueue_render(g, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, stats != nullptr);
    if (rc_q) return rc_q;
    if (stats) memset(stats, 0, sizeof(*stats));
    for (uint32_t r = 0; r < world; ++r) {
        rtx_ctx* c = g->members[r];
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        uint32_t begin = 0, count = 0;
        rtx_spp_range(r, world, spp_begin, spp_count, &begin, &count);
        if (!stats || count == 0) continue;
        double ms = 0.0;
        const int rc = rtx_last_kernel_ms(c, &ms);
        if (rc) return rc;
        if (max_depth == 0 && !(flags & RTX_FLAG_RECURSIVE)) {
            stats->samples += (uint64_t)cols * rows * count;
            stats->term_depth += (uint64_t)cols * rows * count;
        } else {
            unsigned long long h[kStCount + 12];
            HIP_TRY(hipMemcpy(h, c->stats, sizeof(h), hipMemcpyDeviceToHost));
            stats->samples += h[kStSamples]; stats->rays += h[kStRays]; stats->sphere_tests += h[kStRays] * (uint64_t)c->n;
            stats->term_miss += h[kStMiss]; stats->term_absorb += h[kStAbsorb]; stats->term_depth += h[kStDepth];
            stats->candidates += h[kStCandidates]; stats->filter_misses += h[kStFilterMiss];
        }
        stats->kernel_ms = std::max(stats->kernel_ms, ms);       // the members run side by side
    }
    return RTX_OK;
}

// Queues the exchange behind every member's kernels and waits for nothing.  One collective: the sum of rows*cols*3 floats
// (11.5 MB at 1200 x 800, 99.5 MB at 3840 x 2160), in place, each rank's part on that member's own stream.  It is an
// all-reduce, as BASELINE's north star names it, although only member 0 needs the total: RCCL's all-reduce is the collective
// it tunes for the fully connected xGMI node (a reduce to one root is a chain through the ranks), and at these sizes either is
// a fraction of a millisecond.  RTX_GROUP_EXCHANGE=reduce (read per call) sends ncclReduce(root = member 0) instead -- half the
// link traffic -- so that a multi-GPU run can time both (bench.py does, outside its timed region).
static int group_enqueue_exchange(rtx_group* g)
{
    const size_t world = g->members.size();
    rtx_ctx* c0 = g->members[0];
    if (!c0->accum || c0->cols == 0) return fail(RTX_ERR_STATE, "rtx_group_allreduce: nothing rendered yet");
    const size_t count = (size_t)c0->cols * c0->rows * 3;
    for (rtx_ctx* c : g->members)
        if (!c->accum || (size_t)c->cols * c->rows * 3 != count) return fail(RTX_ERR_STATE, "rtx_group_allreduce: the members hold images of different sizes");
    g->ex_timed = false;
    if (world == 1 && g->comms.empty()) return RTX_OK;          // a group of one: the sum over one buffer
    HIP_TRY(hipSetDevice(c0->device));
    if (g->loopback) {
        // member 0's buffer takes the others' in member order, behind everybody's kernels (the members share one device, so
        // waiting for the other members' streams is the cross-stream dependency); then the invariant as below
        for (size_t i = 1; i < world; ++i) HIP_TRY(hipStreamSynchronize(g->members[i]->stream));
        HIP_TRY(hipEventRecord(g->ex0, c0->stream));
        for (size_t i = 1; i < world; ++i) {
            accum_add_kernel<<<(unsigned)((count + 255) / 256), 256, 0, c0->stream>>>(c0->accum, g->members[i]->accum, count);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(g->ex1, c0->stream));
        g->ex_timed = true;
        HIP_TRY(hipStreamSynchronize(c0->stream));
        for (size_t i = 1; i < world; ++i) HIP_TRY(hipMemsetAsync(g->members[i]->accum, 0, count * sizeof(float), g->members[i]->stream));
        return RTX_OK;
    }
    const char* how = getenv("RTX_GROUP_EXCHANGE");
    const bool  to_root = how && strcmp(how, "reduce") == 0;
    HIP_TRY(hipEventRecord(g->ex0, c0->stream));
    NCCL_TRY(g_rccl.GroupStart());
    for (size_t i = 0; i < world; ++i) {
        rtx_ctx* c = g->members[i];
        const ncclResult_t r = to_root ? g_rccl.Reduce(c->accum, c->accum, count, ncclFloat32, ncclSum, 0, g->comms[i], c->stream)
                                       : g_rccl.AllReduce(c->accum, c->accum, count, ncclFloat32, ncclSum, g->comms[i], c->stream);
        if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return fail(RTX_ERR_DEVICE, "the RCCL exchange on member %zu failed: %s", i, g_rccl.GetErrorString(r)); }
    }
    NCCL_TRY(g_rccl.GroupEnd());
    HIP_TRY(hipSetDevice(c0->device));
    HIP_TRY(hipEventRecord(g->ex1, c0->stream));
    g->ex_timed = true;
    // Invariant of a group: the SUM over the members' buffers is the image's colour sum.  After the exchange member 0 holds
    // that total, so all the others start again from zero -- a later RTX_FLAG_ACCUMULATE pass and the next exchange then
    // count every sample once (progressive renders; multi_gpu.render_progressive keeps the same invariant).
    for (size_t i = 1; i < world; ++i) {
        rtx_ctx* c = g->members[i];
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipMemsetAsync(c->accum, 0, count * sizeof(float), c->stream));
    }
    return RTX_OK;
}

static int group_sync(rtx_group* g)
{
    for (rtx_ctx* c : g->members) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return RTX_OK;
}

extern "C" int rtx_group_allreduce(rtx_group* g)
{
    if (!g) return fail(RTX_ERR_ARG, "rtx_group_allreduce: group is NULL");
    const int rc = group_enqueue_exchange(g);
    if (rc) return rc;
    return group_sync(g);
}

// Diagnostic (tools/ only, not declared in rtx.h): host microseconds the most recent rtx_group_render / _render_frame spent queueing the members' launches.
extern "C" double rtx_debug_group_enqueue_us(const rtx_group* g) { return g ? g->last_enqueue_us : 0.0; }

extern "C" int rtx_group_last_exchange_ms(rtx_group* g, double* ms)
{
    if (!g || !ms) return fail(RTX_ERR_ARG, "rtx_group_last_exchange_ms: NULL argument");
    *ms = 0.0;
    if (!g->ex_timed) return RTX_OK;                             // a group of one, or no exchange yet
    HIP_TRY(hipSetDevice(g->members[0]->device));
    HIP_TRY(hipEventSynchronize(g->ex1));
    float f = 0.0f;
    HIP_TRY(hipEventElapsedTime(&f, g->ex0, g->ex1));
    *ms = f;
    return RTX_OK;
}

extern "C" int rtx_group_render_frame(rtx_group* g, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                                      uint64_t seed, uint32_t flags, uint32_t chunk_spp, uint32_t spp_total, uint32_t debug_tile, uint32_t* framebuffer_host)
{
    if (!g || !framebuffer_host) return fail(RTX_ERR_ARG, "rtx_group_render_frame: NULL argument");
    int rc = group_enqueue_render(g, cols, rows, spp_begin, spp_count, max_depth, seed, flags, chunk_spp, false);
    if (rc == RTX_OK) rc = group_enqueue_exchange(g);
    if (rc == RTX_OK) rc = rtx_resolve_rgbx8(g->members[0], spp_total, debug_tile, framebuffer_host);    // waits for member 0's stream
    if (rc) return rc;
    return group_sync(g);                                        // the other members' zero fills
}

extern "C" int rtx_group_resolve_rgbx8(rtx_group* g, uint32_t spp_total, uint32_t debug_tile, uint32_t* framebuffer_host)
{
    if (!g) return fail(RTX_ERR_ARG, "rtx_group_resolve_rgbx8: group is NULL");
    return rtx_resolve_rgbx8(g->members[0], spp_total, debug_tile, framebuffer_host);
}
