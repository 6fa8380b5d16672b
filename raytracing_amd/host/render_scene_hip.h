// render_scene_hip.h -- renderSceneHIP: the fourth backend behind the signature of the reference's renderScene family
// (raytracer.h:17-19), a thin C++ wrapper over the C ABI of include/rtx.h.
//
// Include it AFTER the scene types: either the reference's own scene.h / sphere.h / material.h / camera.h (drop-in
// build) or raytracing_amd/host/scene_types.h (standalone CLI).  It needs, in namespace pk:
// Scene::objects (pointers to IVisible), Sphere{center, radius, material*}, material_t{type, albedo, blur,
// refractionIndex} and Camera{origin, leftCorner, horizontal, vertical, u, v, w, lensRadius}.
//
// Contract (raytracer.cpp:60-143): scene, camera and frameBuffer are borrowed; frameBuffer receives rows*cols pixels
// packed R<<24 | G<<16 | B<<8, row 0 on top; the call blocks until the image is complete; 0 is returned on success
// (the reference always returns 0 -- here a failure prints rtx_last_error() and returns the negative rtx_status).
// numThreads is advisory, as it is for renderSceneCUDA; blockSize only places the -d borders.
#pragma once
#include "../../include/rtx.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace pk
{

// Options beyond the reference's signature (all optional; the defaults give the reference's one-shot behaviour).
struct HipRenderOptions {
    uint64_t    seed = 1;
    int         device = 0;              // first GPU
    int         n_devices = 1;           // GPUs device .. device + n_devices - 1: samples split by index, one RCCL all-reduce per exchange
    int         filter = -1;             // RTX_FILTER_* for the candidate search; -1 keeps the library's default (RTX_FILTER, else RTX_FILTER_AUTO: linear list up to 256 spheres, uniform grid above)
    rtx_stats*  stats = nullptr;         // counters of the last pass
    unsigned    pass_spp = 0;            // > 0: progressive, this many samples per pass
    const char* checkpoint = nullptr;    // written after every pass (rtx_save_accum)
    bool        resume = false;          // continue from `checkpoint` when it exists and matches size, seed, depth and -r
    // called after every pass with the resolved framebuffer and the samples done so far; return false to stop early
    bool (*on_pass)(const uint32_t* frameBuffer, unsigned spp_done, void* user) = nullptr;
    void*       user = nullptr;
};

inline int renderSceneHIP(const Scene& scene, const Camera& camera, unsigned rows, unsigned cols, uint32_t* frameBuffer,
                          unsigned num_aa_samples, unsigned max_ray_depth, unsigned numThreads, unsigned blockSize,
                          bool debug, bool recursive, const HipRenderOptions& opt)
{
    (void)numThreads;
    // the flatten step of raytracer.cpp:76-90, to SoA
    const size_t n = scene.objects.size();
    std::vector<float> cx(n), cy(n), cz(n), rad(n), ar(n), ag(n), ab(n), blur(n), ior(n);
    std::vector<uint32_t> type(n);
    size_t i = 0;
    for (IVisible* obj : scene.objects) {
        const Sphere* s = dynamic_cast<const Sphere*>(obj);
        if (!s || !s->material) { printf("renderSceneHIP: object %zu is not a Sphere with a material\n", i); return RTX_ERR_ARG; }
        cx[i] = s->center.x; cy[i] = s->center.y; cz[i] = s->center.z; rad[i] = s->radius;
        type[i] = (uint32_t)s->material->type;
        ar[i] = s->material->albedo.x; ag[i] = s->material->albedo.y; ab[i] = s->material->albedo.z;
        blur[i] = s->material->blur; ior[i] = s->material->refractionIndex;
        i++;
    }
    rtx_camera cam;
    const vector3* src[7] = { &camera.origin, &camera.leftCorner, &camera.horizontal, &camera.vertical, &camera.u, &camera.v, &camera.w };
    float*         dst[7] = { cam.origin, cam.left_corner, cam.horizontal, cam.vertical, cam.u, cam.v, cam.w };
    for (int k = 0; k < 7; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    cam.lens_radius = camera.lensRadius;

    const uint32_t flags = recursive ? RTX_FLAG_RECURSIVE : 0u;
    // a group of n_devices member contexts; a group of one is a plain context (no RCCL involved)
    std::vector<int> devs;
    // RTX_GROUP_LOOPBACK=1 (rehearsal on a one-GPU box, see rtx_group_create): every member on the first device
    const char* loopback = getenv("RTX_GROUP_LOOPBACK");
    const bool  same_device = loopback && atoi(loopback) != 0;
    for (int d = 0; d < (opt.n_devices > 0 ? opt.n_devices : 1); ++d) devs.push_back(same_device ? opt.device : opt.device + d);
    rtx_group* grp = nullptr;
    int rc = rtx_group_create(devs.data(), (int)devs.size(), &grp);
    rtx_ctx* ctx = rc == RTX_OK ? rtx_group_member(grp, 0) : nullptr;
    for (int d = 0; d < (int)devs.size() && rc == RTX_OK && opt.filter >= 0; ++d) rc = rtx_set_filter(rtx_group_member(grp, d), opt.filter);
    if (rc == RTX_OK) rc = rtx_group_upload_scene(grp, (uint32_t)n, cx.data(), cy.data(), cz.data(), rad.data(), type.data(), ar.data(), ag.data(), ab.data(), blur.data(), ior.data());
    if (rc == RTX_OK) rc = rtx_group_set_camera(grp, &cam);
    unsigned done = 0;
    if (rc == RTX_OK && opt.resume && opt.checkpoint) {
        // the checkpoint is the sum over all devices: it goes to member 0, the others start from zero (first pass below)
        uint32_t c_cols = 0, c_rows = 0, c_done = 0, c_depth = 0, c_flags = 0;
        uint64_t c_seed = 0;
        if (rtx_load_accum(ctx, opt.checkpoint, &c_cols, &c_rows, &c_done, &c_seed, &c_depth, &c_flags) == RTX_OK) {
            if (c_cols == cols && c_rows == rows && c_seed == opt.seed && c_depth == max_ray_depth && c_flags == flags && c_done <= num_aa_samples) {
                done = c_done;
                printf("renderSceneHIP: resuming %s at %u of %u samples\n", opt.checkpoint, done, num_aa_samples);
                if (devs.size() > 1) {
                    std::vector<float> zeros((size_t)cols * rows * 3, 0.0f);
                    for (int d = 1; d < (int)devs.size() && rc == RTX_OK; ++d) rc = rtx_write_accum(rtx_group_member(grp, d), cols, rows, zeros.data(), zeros.size());
                }
            } else {
                printf("renderSceneHIP: %s belongs to another render (size, seed, depth or -r differ); starting over\n", opt.checkpoint);
            }
        }
    }
    const unsigned pass = opt.pass_spp ? opt.pass_spp : num_aa_samples;
    bool go_on = true;
    while (rc == RTX_OK && go_on && done < num_aa_samples) {
        const unsigned count = (num_aa_samples - done < pass) ? num_aa_samples - done : pass;
        rc = rtx_group_render(grp, cols, rows, done, count, max_ray_depth, opt.seed, flags | (done ? RTX_FLAG_ACCUMULATE : 0u), 0, opt.stats);
        if (rc != RTX_OK) break;
        done += count;
        const bool show = opt.on_pass || done == num_aa_samples;
        if (opt.checkpoint || show) rc = rtx_group_allreduce(grp);        // member 0 now holds the sum over the devices
        if (rc == RTX_OK && opt.checkpoint) rc = rtx_save_accum(ctx, opt.checkpoint, done, opt.seed, max_ray_depth, flags);
        if (rc == RTX_OK && show) {
            rc = rtx_group_resolve_rgbx8(grp, done, debug ? blockSize : 0u, frameBuffer);
            if (rc == RTX_OK && opt.on_pass) go_on = opt.on_pass(frameBuffer, done, opt.user);
        }
    }
    if (rc == RTX_OK && done == num_aa_samples && opt.resume) rc = rtx_group_resolve_rgbx8(grp, done, debug ? blockSize : 0u, frameBuffer);   // resumed a finished render
    if (rc != RTX_OK) printf("renderSceneHIP: %s\n", rtx_last_error());
    rtx_group_destroy(grp);
    return rc;
}

// The reference's signature (raytracer.h:17-19) plus trailing defaults.
inline int renderSceneHIP(const Scene& scene, const Camera& camera, unsigned rows, unsigned cols, uint32_t* frameBuffer,
                          unsigned num_aa_samples = 4, unsigned max_ray_depth = 50, unsigned numThreads = 1, unsigned blockSize = 64,
                          bool debug = false, bool recursive = true, uint64_t seed = 1, int device = 0, rtx_stats* stats = nullptr, int n_devices = 1)
{
    HipRenderOptions opt;
    opt.seed = seed; opt.device = device; opt.stats = stats; opt.n_devices = n_devices;
    return renderSceneHIP(scene, camera, rows, cols, frameBuffer, num_aa_samples, max_ray_depth, numThreads, blockSize, debug, recursive, opt);
}

} // namespace pk
