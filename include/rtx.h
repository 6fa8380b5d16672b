/*
 * rtx.h -- C ABI of the MI355X (gfx950) render backend for the per-pixel trace loop of avneic/RayTracing.
 *
 * The library behind this header (raytracing_amd/csrc -> librtx.so) replaces, for the path
 *   renderScene -> _renderJob -> (Camera::getRay, _color -> _sceneHit -> sphereHit, materialScatter)
 * what the reference implements three times behind one signature (raytracer.h:17-19: renderScene,
 * renderSceneCUDA, renderSceneISPC).  It is the fourth backend: `renderSceneHIP` (raytracing_amd/host/
 * render_scene_hip.h) has that same signature and is a thin C++ wrapper over the calls below, the way
 * renderSceneISPC is a wrapper over the extern "C" pair cameraInitISPC / renderISPC (raytracer_ispc.h:93-95)
 * that take one POD context (raytracer_ispc.h:44-64).
 *
 * Conventions: every call returns 0 on success and a negative rtx_status on failure; rtx_last_error() returns
 * the message of the last failure on the calling thread.  All calls block until their work is complete.  A
 * context belongs to one device and is not thread-safe.  There is NO CPU fallback: without a usable gfx950
 * device rtx_create fails and nothing else can be called.
 *
 * Pointers named *_host are host memory, *_dev are device memory of the context's GPU.  Plain pointers and
 * sizes only -- no C++ or torch types cross this boundary.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RTX_OK            = 0,
    RTX_ERR_ARG       = -1,  /* bad argument (null pointer, zero size, unknown material type, ...) */
    RTX_ERR_DEVICE    = -2,  /* no gfx950 device / HIP runtime error */
    RTX_ERR_STATE     = -3,  /* call order (render before scene/camera upload, ...) */
    RTX_ERR_NOMEM     = -4
} rtx_status;

/* material.h:9-14 (material_type_t) */
enum { RTX_MATERIAL_NONE = 0, RTX_MATERIAL_DIFFUSE = 1, RTX_MATERIAL_METAL = 2, RTX_MATERIAL_GLASS = 3 };

/* The derived camera of camera.h:94-101 (what Camera's constructor computes, camera.h:20-41). */
typedef struct rtx_camera {
    float origin[3];
    float left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3];
    float v[3];
    float w[3];
    float lens_radius;
} rtx_camera;

/* rtx_render flags */
enum {
    RTX_FLAG_ACCUMULATE = 1u << 0,  /* add this call's colour sums to the accumulation buffer instead of replacing it */
    RTX_FLAG_RECURSIVE  = 1u << 1,  /* the -r semantics of raytracer.cpp:213-240: absorbed / over-depth paths are black, the ray at
                                       depth == max_depth is still traced, attenuations are multiplied innermost first (needs a
                                       per-lane stack of max_depth float4 in device memory, at most 8 GiB in all) */
    RTX_FLAG_VALIDATE_FILTER = 1u << 2 /* debug: also run the reference's full ascending scan (raytracer.cpp:291-300) for
                                       every ray and count disagreements with the filtered result in stats.filter_misses */
};

/* Candidate filters in front of the exact sphereHit arithmetic (results are identical; only speed differs). */
enum { RTX_FILTER_VALU = 0, RTX_FILTER_MFMA = 1,
       RTX_FILTER_GRID = 2   /* an acceleration structure instead of a filter over the whole list: a uniform grid
                                over the table, walked per ray; spheres too large for it are tested for every ray.  For large
                                tables (faster than the MFMA filter from about 300 spheres); the linear list of the
                                reference is RTX_FILTER_MFMA */,
       RTX_FILTER_AUTO = 3   /* RTX_FILTER_MFMA up to 256 spheres, RTX_FILTER_GRID from there on */ };

/* Counters gathered inside the kernel with wave ballots (rays = _sceneHit calls, raytracer.cpp:251). */
typedef struct rtx_stats {
    uint64_t samples;
    uint64_t rays;
    uint64_t sphere_tests;      /* rays * n: every ray is tested against the whole list, as in raytracer.cpp:291-300 */
    uint64_t term_miss;         /* paths ended on the background (raytracer.cpp:266-269) */
    uint64_t term_absorb;       /* paths ended by materialScatter returning false (raytracer.cpp:262-264) */
    uint64_t term_depth;        /* paths that used up max_depth bounces (raytracer.cpp:250) */
    uint64_t candidates;        /* (ray, sphere) pairs that went through the exact sphereHit arithmetic */
    uint64_t filter_misses;     /* RTX_FLAG_VALIDATE_FILTER only: rays whose filtered hit differs from the full scan (must be 0) */
    double   kernel_ms;         /* trace + reduce kernels, HIP events on the context's stream */
} rtx_stats;

typedef struct rtx_ctx rtx_ctx;

/* ---- lifetime ------------------------------------------------------------------------------------------ */

/* Replaces the per-call pool/scene set-up of raytracer.cpp:62-90 (and raytracer.cu:57-109): binds a context to
 * HIP device `device_id`, creates its stream.  Fails with RTX_ERR_DEVICE when the device is absent or not gfx950. */
int rtx_create(int device_id, rtx_ctx** out_ctx);
int rtx_destroy(rtx_ctx* ctx);
const char* rtx_last_error(void);

/* ---- inputs -------------------------------------------------------------------------------------------- */

/* Replaces the AoS flatten of raytracer.cpp:76-90 / raytracer.cu:77-94: n spheres as SoA host arrays (copied).
 * mat_type[i] in {DIFFUSE, METAL, GLASS}; albedo, blur, ior as material.h:17-38.  n may be 0. */
int rtx_upload_scene(rtx_ctx* ctx, uint32_t n,
                     const float* cx_host, const float* cy_host, const float* cz_host, const float* radius_host,
                     const uint32_t* mat_type_host,
                     const float* albedo_r_host, const float* albedo_g_host, const float* albedo_b_host,
                     const float* blur_host, const float* ior_host);

/* Chooses the candidate search (default RTX_FILTER_AUTO; the environment variable RTX_FILTER=valu|mfma|grid|auto overrides the
 * default at rtx_create).  No counterpart in the reference: all forms feed the same exact sphere.cpp:68-102 test and keep the
 * minimum of (t, index), i.e. the winner of the linear loop of raytracer.cpp:285-304, so the images are identical bit for bit;
 * RTX_FILTER_MFMA is that linear loop over the whole list (what bench.py's headline measures), RTX_FILTER_GRID visits only the
 * spheres near the ray. */
int rtx_set_filter(rtx_ctx* ctx, int kind);

/* Tuning knobs of the launch (no counterpart in the reference; results do not depend on them except through the order
 * in which a pixel's sample chunks are summed).  Each knob's initial value comes from the environment variable named
 * beside it, read once at rtx_create.  Unknown keys and out-of-range values fail with RTX_ERR_ARG.
 *   "block"          RTX_BLOCK           workgroup size, a multiple of 64 up to 512; 0 = automatic
 *   "stream"         RTX_STREAM          stream the MFMA filter's table through the LDS stage: -1 automatic, 0 never, 1 always
 *   "items_per_lane" RTX_ITEMS_PER_LANE  work items per lane that the smallest sample chunks are sized for (1024)
 *   "max_chunks"     RTX_MAX_CHUNKS      most sample chunks of an automatic plan (63)
 *   "tail_factor"    RTX_TAIL_FACTOR     how many mean item durations the plan allows a single item to take (30)
 *   "partials_mb"    RTX_PARTIALS_MB     cap of the partial-sum scratch (16 B per chunk and pixel) in MiB (1024)
 *   "wave_batch"     RTX_WAVE_BATCH      work items a wave reserves per trip to the counter; 0 = automatic
 *   "lds_pad"        RTX_LDS_PAD         extra LDS bytes per workgroup (experiments on occupancy)
 *   "grid_density_pct" RTX_GRID_DENSITY_PCT  RTX_FILTER_GRID: spheres per grid cell x 100 (35); takes effect at the next rtx_upload_scene
 *   "order"          RTX_ORDER           1 (default): a launch of a frame the context has traced before (same scene, camera, size;
 *                                        automatic chunks) hands its work out most expensive strip of 64 pixels first, by the
 *                                        rays the strips took in the previous launch; 0: never
 *   "order_chunks"   RTX_ORDER_CHUNKS    most sample chunks of such a launch (16)
 *   "order_share_pct" RTX_ORDER_SHARE_PCT share of such a launch's expected duration that its longest work item may take (50)
 *   "tail_guard"     RTX_TAIL_GUARD      the last tail_guard x resident lanes work items are reserved on demand, not ahead (4)
 *   "walk_park"      RTX_WALK_PARK       grid filter: the walks of a wave's last walk_park rays are parked in LDS while the wave shades
 *                                        the others, and go on in the next iteration (10; 0: every walk ends in its own iteration; at most 32)
 *   "print_phases"   RTX_PRINT_PHASES    diagnostic builds only */
int rtx_set_tuning(rtx_ctx* ctx, const char* key, int value);

/* Replaces raytracer.cu:68-73 (_createCamera) / raytracerispc.cpp:91-103 (cameraInitISPC). */
int rtx_set_camera(rtx_ctx* ctx, const rtx_camera* cam);

/* ---- the hot path ---------------------------------------------------------------------------------------- */

/* Replaces the tile fan-out and _renderJob's sample loop (raytracer.cpp:93-133, :145-186; raytracer.cu:112-114):
 * traces samples [spp_begin, spp_begin + spp_count) of every pixel of a cols x rows image and leaves the per-pixel
 * colour SUMS (not yet divided by spp) in the context's float accumulation buffer, rows*cols*3 floats, row-major,
 * row 0 on top, RGB interleaved.
 *   seed       : key of the Philox4x32-10 sample stream ("random stream")
 *   chunk_spp  : samples summed per work item before the ordered chunk reduction; 0 = choose automatically
 *   primary_hit_host : NULL, or spp_count*rows*cols uint32 that receive the sphere index of each sample's first
 *                hit, [(s - spp_begin)*rows*cols + y*cols + x], 0xFFFFFFFF on a miss
 *   stats      : NULL or counters of this call */
int rtx_render(rtx_ctx* ctx, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
               uint64_t seed, uint32_t flags, uint32_t chunk_spp, uint32_t* primary_hit_host, rtx_stats* stats);

/* The sample chunks of the most recent rtx_render / rtx_render_async (no counterpart in the reference): a pixel's samples
 * are summed per chunk, in sample order, and the chunk sums are added in chunk order.  *chunk_spp receives the uniform chunk
 * size when the caller chose one; otherwise it receives 0, *n_chunks the number of chunks of the automatic plan (large
 * chunks first, small ones last) and starts[0 .. *n_chunks] their boundaries relative to spp_begin (`cap` entries are
 * available; 64 always suffice).  A CPU restatement summing in that same order reproduces an automatic launch bit for bit. */
int rtx_last_chunk_plan(rtx_ctx* ctx, uint32_t* chunk_spp, uint32_t* n_chunks, uint32_t* starts, uint32_t cap);

/* Whether the most recent rtx_render / rtx_render_async ran a cost-ordered work list (no counterpart in the reference, whose
 * thread pool hands tiles out in image order, raytracer.cpp:93-127): the first launch of a frame (scene, camera, size) measures
 * what every strip of 64 consecutive pixels costs -- the most rays one of its pixels took -- and later launches of that frame
 * start the expensive strips first, so that what is still running when the work list is exhausted are the cheapest items of
 * the frame.  Results never depend on the order (a pixel's samples are summed per chunk, the chunks in chunk order);
 * rtx_last_chunk_plan reports the chunks either way. */
int rtx_last_launch_ordered(rtx_ctx* ctx, int* ordered);

/* rtx_render for callers that cannot allocate: it calls no hipMalloc / hipFree and no routine that allocates.  It has no
 * counterpart in the reference; it is for callers that must not allocate while they render -- tight serving loops, harnesses
 * that forbid it, memory-accounted processes.  (It is NOT capturable into a HIP graph: it runs on the context's own stream,
 * queries the pointer and blocks until the image is complete.)  Inputs are the scene and camera already uploaded to `ctx`; the
 * output is accum_dev, cols*rows*3 floats of the CALLER's device memory on the context's GPU (colour sums as for rtx_render;
 * RTX_FLAG_ACCUMULATE adds to what it holds; an allocation smaller than that is refused with RTX_ERR_ARG).  The work counter and
 * the partial sums live in a static device array of 64 MiB + 256 bytes that every GPU the library runs on holds until the
 * process ends: a launch may use at most 4 194 304 work items (pixels x sample chunks; automatic plans shrink to fit, which
 * lengthens the tail of large frames), more pixels than that fail with RTX_ERR_ARG.  RTX_FLAG_RECURSIVE (needs a per-lane
 * stack) is refused; no primary-hit map, no statistics, no cost-ordered work list.  All calls on one GPU share that array:
 * calls that overlap on a GPU -- from two contexts or two threads -- would race, so the array's first word is an "in use" flag
 * (a host-side atomic per device) and the second of two overlapping calls of one process fails with RTX_ERR_STATE instead. */
int rtx_render_noalloc(rtx_ctx* ctx, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                       uint64_t seed, uint32_t flags, uint32_t chunk_spp, float* accum_dev);

/* Same, but only enqueues the work on the context's stream and returns; rtx_sync() waits.  bench.py brackets
 * K of these with one rtx_sync.  Statistics and primary-hit maps are not available in this form. */
int rtx_render_async(rtx_ctx* ctx, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                     uint64_t seed, uint32_t flags, uint32_t chunk_spp);
int rtx_sync(rtx_ctx* ctx);
/* Milliseconds between the first and the last kernel enqueued since the previous rtx_sync (HIP events). */
int rtx_last_kernel_ms(rtx_ctx* ctx, double* ms);

/* Milliseconds the most recent trace kernel alone took (HIP events on the context's stream, around that launch). */
int rtx_last_trace_ms(rtx_ctx* ctx, double* ms);

/* ---- outputs --------------------------------------------------------------------------------------------- */

/* The accumulation buffer in device memory (valid until the next rtx_render with another image size or
 * rtx_destroy).  The multi-GPU host sums these across ranks in place (RCCL all-reduce) before resolving. */
int rtx_accum_device(rtx_ctx* ctx, float** accum_dev, size_t* n_floats);
/* Makes the context accumulate into caller-owned device memory (cols*rows*3 floats on the context's GPU, e.g. a
 * torch tensor's data_ptr) so that the host can hand the very same buffer to its collective library.  NULL returns
 * to a library-owned buffer.  The reference's caller owns the framebuffer in the same way (RayTracing.cpp:123-129). */
int rtx_use_accum_buffer(rtx_ctx* ctx, uint32_t cols, uint32_t rows, float* accum_dev);
int rtx_read_accum(rtx_ctx* ctx, float* rgb_host, size_t n_floats);
int rtx_write_accum(rtx_ctx* ctx, uint32_t cols, uint32_t rows, const float* rgb_host, size_t n_floats);

/* Progressive / resumable rendering (no counterpart in the reference, whose renderScene is one
 * blocking call): a checkpoint is the accumulation buffer plus what is needed to continue it -- image size, samples
 * done so far, seed, max_depth and flags.  rtx_save_accum writes a 48-byte header ("RTXACCUM", then little-endian u32
 * version = 1, cols, rows, spp_done, u64 seed, u32 max_depth, flags, two reserved zero words), then rows*cols*3 floats; rtx_load_accum reads one back into the context's accumulation buffer (as rtx_write_accum
 * does) and returns the fields; the caller continues with rtx_render(spp_begin = *spp_done, RTX_FLAG_ACCUMULATE). */
int rtx_save_accum(rtx_ctx* ctx, const char* path, uint32_t spp_done, uint64_t seed, uint32_t max_depth, uint32_t flags);
int rtx_load_accum(rtx_ctx* ctx, const char* path, uint32_t* cols, uint32_t* rows, uint32_t* spp_done, uint64_t* seed,
                   uint32_t* max_depth, uint32_t* flags);

/* Replaces raytracer.cpp:187-197: colour sum * (1/spp_total), sqrt (gamma 2), (uint8_t)(255.99 * c) in double,
 * pack R<<24 | G<<16 | B<<8 into the caller's framebuffer (rows*cols uint32, the layout RayTracing.cpp:156-165
 * unpacks).  debug_tile != 0 paints tile borders 0xFF000000 as -d does (raytracer.cpp:168-171). */
int rtx_resolve_rgbx8(rtx_ctx* ctx, uint32_t spp_total, uint32_t debug_tile, uint32_t* framebuffer_host);

/* Host memory for framebuffers, pinned and mapped into the GPU (hipHostMalloc, portable + mapped): an rtx_resolve_rgbx8 whose
 * framebuffer_host lies inside such a block is written by the resolve kernel itself over the host link -- no staging copy
 * (0.36 -> about 0.1 ms per 1200 x 800 frame).  The reference's CUDA mode allocates its framebuffer through the GPU runtime in
 * the same way (cudaMallocManaged, RayTracing.cpp:125; new[] otherwise, :127).  rtx_host_free or rtx_destroy releases the block;
 * any other host memory keeps working as before, through a copy.  For a group: allocate through rtx_group_member(group, 0), the
 * member that resolves.  (Pinning memory the caller already owns -- hipHostRegister -- is refused by the runtime on the
 * machines this was built on, hence an allocator.) */
int rtx_host_alloc(rtx_ctx* ctx, size_t bytes, void** host);
int rtx_host_free(rtx_ctx* ctx, void* host);

/* ---- several GPUs of one node (no counterpart in the reference, which is single-device) -------- */

/* A group is one member context per listed device, in ONE process, plus -- for more than one device -- a RCCL communicator
 * over them (ncclCommInitAll; xGMI inside a node).  The path shards by sample index: member r of R traces the global
 * samples rtx_spp_range(r, R, ...) of EVERY pixel into its own full-frame accumulation buffer; the Philox counter carries
 * the global sample index, so the union over members is exactly the one-GPU sample set and primary hits do not depend
 * on the device count.  rtx_group_allreduce is the one exchange: ncclAllReduce(sum, float32) of rows*cols*3 floats in
 * place (RTX_GROUP_EXCHANGE=reduce in the environment: ncclReduce to member 0 instead).  Invariant: the SUM over the members' buffers is the image's colour sum -- after the exchange member 0 holds the
 * total and the others are zeroed, so RTX_FLAG_ACCUMULATE passes and further exchanges count every sample once.
 * A group of one device is rtx_render bit for bit (its exchange is the identity and RCCL is not initialised).
 * All calls block.  Members can be tuned individually through rtx_group_member.  Rehearsal on a one-GPU box: with
 * RTX_GROUP_LOOPBACK=1 in the environment the ids may all name the same device and the exchange is a plain sum into member 0. */
typedef struct rtx_group rtx_group;
int rtx_group_create(const int* device_ids, int n_devices, rtx_group** out_group);
int rtx_group_destroy(rtx_group* group);
int rtx_group_size(const rtx_group* group);
rtx_ctx* rtx_group_member(rtx_group* group, int rank);
int rtx_group_upload_scene(rtx_group* group, uint32_t n,
                           const float* cx_host, const float* cy_host, const float* cz_host, const float* radius_host,
                           const uint32_t* mat_type_host,
                           const float* albedo_r_host, const float* albedo_g_host, const float* albedo_b_host,
                           const float* blur_host, const float* ior_host);
int rtx_group_set_camera(rtx_group* group, const rtx_camera* cam);
/* rtx_render over the group: every member traces its share, all devices side by side.  stats (may be NULL) are summed
 * over the members, kernel_ms is the slowest member's. */
int rtx_group_render(rtx_group* group, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                     uint64_t seed, uint32_t flags, uint32_t chunk_spp, rtx_stats* stats);
int rtx_group_allreduce(rtx_group* group);
/* Milliseconds that member 0's part of the most recent exchange took (HIP events on member 0's stream around the RCCL group
 * call; 0 for a group of one).  When the members were not synchronised beforehand it includes the wait for the slowest member. */
int rtx_group_last_exchange_ms(rtx_group* group, double* ms);
/* One whole frame pass without a host wait in between: rtx_group_render (no statistics) + rtx_group_allreduce +
 * rtx_group_resolve_rgbx8(spp_total), each queued behind the other on the members' streams -- what the body of a multi-GPU
 * renderScene (raytracer.cpp:60-143) amounts to; returns when framebuffer_host is filled. */
int rtx_group_render_frame(rtx_group* group, uint32_t cols, uint32_t rows, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                           uint64_t seed, uint32_t flags, uint32_t chunk_spp, uint32_t spp_total, uint32_t debug_tile, uint32_t* framebuffer_host);
/* rtx_resolve_rgbx8 on member 0 (call rtx_group_allreduce first). */
int rtx_group_resolve_rgbx8(rtx_group* group, uint32_t spp_total, uint32_t debug_tile, uint32_t* framebuffer_host);
/* Contiguous share of [spp_begin, spp_begin + spp_count) of member `rank` of `world`; remainder samples go to the lowest
 * ranks.  Host arithmetic only: the one-process-per-GPU host (raytracing_amd/multi_gpu.py) splits by the same rule. */
void rtx_spp_range(uint32_t rank, uint32_t world, uint32_t spp_begin, uint32_t spp_count, uint32_t* begin, uint32_t* count);

/* ---- host-side helpers (no GPU needed) ---------------------------------------------------------------------- */

/* camera.h:20-41: derives the camera from the constructor's arguments, in the constructor's operation order. */
int rtx_camera_init(rtx_camera* cam, float vfov, float aspect, float aperture, float focus_distance,
                    const float pos[3], const float up[3], const float lookat[3]);

/* RayTracing.cpp:182-213 (_randomScene), driven by a seeded Philox stream instead of the unseeded mt19937.
 * grid_half = 11, p_diffuse = 0.8, p_metal = 0.95 is the book-1 final scene.  Arrays must hold
 * rtx_scene_capacity(grid_half) entries; *n receives the number of spheres written. */
uint32_t rtx_scene_capacity(int grid_half);
int rtx_random_scene(uint64_t seed, int grid_half, float p_diffuse, float p_metal, uint32_t* n,
                     float* cx, float* cy, float* cz, float* radius, uint32_t* mat_type,
                     float* albedo_r, float* albedo_g, float* albedo_b, float* blur, float* ior);

/* The automatic sample-chunk plan of rtx_render as a pure function (no GPU, no counterpart in the reference): fills
 * starts[0 .. *n_chunks] (cap >= 64) for `spp_count` samples of `n_pixels` pixels traced by `lanes` resident lanes, with
 * the knobs of rtx_set_tuning and at most max_items = chunks * pixels work items.  Large chunks come first and shrink
 * geometrically to the size that gives every lane `items_per_lane` items: only the END of a launch needs small items. */
int rtx_plan_chunks(uint32_t spp_count, uint64_t n_pixels, uint64_t lanes, uint32_t items_per_lane, uint32_t max_chunks,
                    uint32_t tail_factor, uint64_t max_items, uint32_t* starts, uint32_t cap, uint32_t* n_chunks);

/* The uniform grid of RTX_FILTER_GRID as a pure function (no GPU, no counterpart in the reference; for tests and inspection):
 * what rtx_upload_scene builds for these spheres at `density_pct` spheres per cell x 100.  info is always filled; the arrays
 * are filled when non-NULL and large enough (cell_start: ncells + 1 entries with ncells = dim[0]*dim[1]*dim[2], items: n_items,
 * big: n_big, tier_end: 3 * ncells), else RTX_ERR_ARG.  Cell (x, y, z) = index c = (z*dim[1] + y)*dim[0] + x holds the spheres
 * items[cell_start[c] .. cell_start[c+1]) in three parts that end at tier_end[3c], tier_end[3c+1], tier_end[3c+2]: tier k
 * registers a sphere wherever its bounding box grown by pad + blur(r, tier_l[k]) reaches (the reference's fp32 sphere test
 * accepts rays that pass a sphere at up to blur(r, L) = sqrt(r^2 + 2^-19 (L^2 + r^2)) - r beyond its surface when the ray's
 * origin is L away); a ray reads the parts 0..k of its cells when it cannot get farther than tier_s[k] (k = 0, 1) from its
 * origin inside the grid; tier_s[2] = tier_l[2]: when the origin's distance from the box + box_diag exceeds it, the ray is clipped
 * against the box grown by pad + blur(r_min, that distance) and, if it enters, scans the table instead. */
typedef struct rtx_grid_info {
    float    gmin[3], inv_h[3], h[3];
    uint32_t dim[3];
    float    pad;
    uint32_t n_big;
    uint64_t n_items;
    float    tier_s[3];
    double   tier_l[3], tier_grow[3];
    float    box_diag, r_min;      /* diagonal of the grid box; smallest binned radius */
} rtx_grid_info;
int rtx_build_grid(uint32_t n, const float* cx, const float* cy, const float* cz, const float* radius, uint32_t density_pct,
                   rtx_grid_info* info, uint32_t* cell_start, uint64_t cell_start_cap, uint32_t* items, uint64_t items_cap,
                   uint32_t* big, uint64_t big_cap, uint32_t* tier_end, uint64_t tier_end_cap);

/* RayTracing.cpp:152-165: "P3\n<cols> <rows>\n255\n" then "r g b\n" per pixel. */
int rtx_write_ppm(const char* path, uint32_t cols, uint32_t rows, const uint32_t* framebuffer_host);
/* The same image as binary "P6" (3 bytes per pixel): 4x smaller and much faster to write at 4K; P3 stays the default. */
int rtx_write_ppm_p6(const char* path, uint32_t cols, uint32_t rows, const uint32_t* framebuffer_host);

const char* rtx_version(void);

#ifdef __cplusplus
}
#endif
#endif
