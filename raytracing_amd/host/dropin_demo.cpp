// dropin_demo.cpp -- the drop-in claim, end to end: a main() around renderSceneHIP, wired exactly as INTEGRATION.md section 2
// tells a maintainer to wire the fourth backend in.  One source, two builds:
//   * the reference-typed build (the oracle's Makefile, target `ref`): against THE REFERENCE'S OWN scene.h / sphere.h / material.h /
//     camera.h and its sphere.cpp.  Exists only where the reference tree does and never leaves that container: there it
//     proves that the wrapper compiles and links against the reference's types and fails loudly without a GPU
//     (tests/test_host_cabi.py).
//   * raytracing_amd/dropin_twin (csrc/Makefile, target `cli`, -DDROPIN_OWN_TYPES): the same main() against
//     host/scene_types.h, the own types with the same names and fields.  This one runs on the GPU box, where
//     tests/test_gpu_dropin.py compares its framebuffer with the golden frames.
//
//   dropin_demo <scene.bin> <out.fb.u32> <cols> <rows> <spp> <max_depth> <recursive>
// scene.bin: uint32 n, then n records of 10 float32: cx cy cz radius type albedo_r albedo_g albedo_b blur ior.
#ifdef DROPIN_OWN_TYPES
#include "scene_types.h"
#else
#include "camera.h"
#include "material.h"
#include "scene.h"
#include "sphere.h"
#endif

#include "render_scene_hip.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace pk;

int main(int argc, char** argv)
{
    if (argc < 8) { fprintf(stderr, "usage: dropin_demo <scene.bin> <out.fb.u32> <cols> <rows> <spp> <max_depth> <recursive>\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return 2;
    std::vector<float> rec((size_t)n * 10);
    if (fread(rec.data(), 4, rec.size(), f) != rec.size()) return 2;
    fclose(f);
    Scene scene;                                                        // scene.h:34-47, filled as RayTracing.cpp:182-213 does
    for (uint32_t i = 0; i < n; ++i) {
        const float* r = &rec[(size_t)i * 10];
        scene.objects.push_back(new Sphere(vector3(r[0], r[1], r[2]), r[3], new material_t((material_type_t)(int)r[4], vector3(r[5], r[6], r[7]), r[8], r[9])));
    }
    const unsigned cols = atoi(argv[3]), rows = atoi(argv[4]), spp = atoi(argv[5]), depth = atoi(argv[6]);
    const bool recursive = atoi(argv[7]) != 0;
    Camera camera(20, float(cols) / float(rows), 0.1f, 10.0f, vector3(13, 2, 3), vector3(0, 1, 0), vector3(0, 0, 0));   // RayTracing.cpp:107-117
    std::vector<uint32_t> frameBuffer((size_t)rows * cols, 0xCCCCCCCCu);
    const int rc = renderSceneHIP(scene, camera, rows, cols, frameBuffer.data(), spp, depth, 1, 64, false, recursive);
    if (rc != 0) return 1;
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    fwrite(frameBuffer.data(), 4, frameBuffer.size(), o);
    fclose(o);
    return 0;
}
