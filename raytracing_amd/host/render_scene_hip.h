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
#include <vector>

namespace pk
{

inline int renderSceneHIP(const Scene& scene, const Camera& camera, unsigned rows, unsigned cols, uint32_t* frameBuffer,
                          unsigned num_aa_samples = 4, unsigned max_ray_depth = 50, unsigned numThreads = 1, unsigned blockSize = 64,
                          bool debug = false, bool recursive = true, uint64_t seed = 1, int device = 0, rtx_stats* stats = nullptr)
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

    rtx_ctx* ctx = nullptr;
    int rc = rtx_create(device, &ctx);
    if (rc == RTX_OK) rc = rtx_upload_scene(ctx, (uint32_t)n, cx.data(), cy.data(), cz.data(), rad.data(), type.data(), ar.data(), ag.data(), ab.data(), blur.data(), ior.data());
    if (rc == RTX_OK) rc = rtx_set_camera(ctx, &cam);
    if (rc == RTX_OK) rc = rtx_render(ctx, cols, rows, 0, num_aa_samples, max_ray_depth, seed, recursive ? RTX_FLAG_RECURSIVE : 0u, 0, nullptr, stats);
    if (rc == RTX_OK) rc = rtx_resolve_rgbx8(ctx, num_aa_samples, debug ? blockSize : 0u, frameBuffer);
    if (rc != RTX_OK) printf("renderSceneHIP: %s\n", rtx_last_error());
    rtx_destroy(ctx);
    return rc;
}

} // namespace pk
