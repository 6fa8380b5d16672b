// ref_harness.cpp -- drives the COMPILED REFERENCE (avneic/RayTracing, /root/reference) to produce golden vectors.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ref_harness (git-ignored); it never ships
// in the product and nothing under raytracing_amd/ or include/ refers to it.
//
// The reference translation units are #included from where they lie so that their file-static functions
// (_color, _sceneHit, _background, _renderJob in raytracer.cpp:53-57; _reflect, _refract, _schlick in
// material.cu:9-15) can be called directly.  No reference source is copied into this repository.
//
//   ref_harness sizes
//   ref_harness kat <outdir>
//   ref_harness frame <scene.bin> <out_prefix> <cols> <rows> <spp_begin> <spp_count> <max_depth> <seed> <recursive>
//   ref_harness render_ref <scene.bin> <out.fb.u32> <cols> <rows> <spp> <max_depth> <threads> <block> <recursive> <mt_seed>
//   ref_harness scene_roundtrip <scene.bin> <out.bin>
//
// scene.bin: uint32 n, then n records of 10 float32: cx cy cz radius type albedo_r albedo_g albedo_b blur ior.
#include "ref_inject.h"

#include REF_RAYTRACER_CPP
#include REF_MATERIAL_CU
#include REF_SPHERE_CPP

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

using namespace pk;

static std::vector<float> g_row;
static void put(float v) { g_row.push_back(v); }
static void put(const vector3& v) { put(v.x); put(v.y); put(v.z); }
static void put_u32(uint32_t v) { float f; memcpy(&f, &v, 4); put(f); }

struct Writer {
    FILE* f;
    explicit Writer(const std::string& path) : f(fopen(path.c_str(), "wb")) { if (!f) { perror(path.c_str()); exit(1); } }
    ~Writer() { fclose(f); }
    void row() { fwrite(g_row.data(), sizeof(float), g_row.size(), f); g_row.clear(); }
};

static std::mt19937                          g_gen(20261110u);
static std::uniform_real_distribution<float> g_uni(0.0f, 1.0f);
static float   U() { return g_uni(g_gen); }
static float   S() { return 2.0f * U() - 1.0f; }
static vector3 randdir()
{
    vector3 d;
    do { d = vector3(S(), S(), S()); } while (d.squared_length() < 1e-3f);
    return d;
}

static std::vector<sphere_t> load_scene(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) exit(1);
    std::vector<sphere_t> out(n);
    for (uint32_t i = 0; i < n; ++i) {
        float r[10];
        if (fread(r, 4, 10, f) != 10) exit(1);
        out[i].center   = vector3(r[0], r[1], r[2]);
        out[i].radius   = r[3];
        out[i].material = material_t((material_type_t)(int)r[4], vector3(r[5], r[6], r[7]), r[8], r[9]);
    }
    fclose(f);
    return out;
}

static Camera book_camera(unsigned cols, unsigned rows)
{
    // RayTracing.cpp:107-117
    vector3 origin(13, 2, 3), lookat(0, 0, 0), up(0, 1, 0);
    return Camera(20, float(cols) / float(rows), 0.1f, 10.0f, origin, up, lookat);
}

// ---------------------------------------------------------------------------------------------------------------
static void kat_sphere_hit(const std::string& dir)
{
    Writer w(dir + "/kat_sphere_hit.f32");
    for (int i = 0; i < 2400; ++i) {
        sphere_t s;
        int      kind = i % 8;
        s.center      = vector3(10 * S(), 10 * S(), 10 * S());
        s.radius      = (kind == 6) ? 1000.0f : 0.05f + 2.0f * U();
        if (kind == 6) s.center = vector3(0, -1000.0f, 0);
        s.material    = material_t(MATERIAL_DIFFUSE, vector3(U(), U(), U()), U(), 1.5f);
        ray r;
        float tmin = 0.001f, tmax = FLT_MAX;
        vector3 o(20 * S(), 20 * S(), 20 * S());
        if (kind == 6) o = vector3(15 * S(), 3 * U(), 15 * S());
        vector3 target = s.center + (s.radius * (kind == 1 ? 3.0f : 0.9f)) * vector3(S(), S(), S());
        vector3 d      = (target - o) * (0.1f + 3.0f * U());
        if (kind == 2) { // graze: aim at a point on the silhouette, nudged by a few ulps
            vector3 oc = s.center - o;
            vector3 t  = oc.cross(randdir()).normalized();
            float   k  = 1.0f + (float)((i / 8) % 9 - 4) * 1.2e-7f;
            d          = (s.center + (s.radius * k) * t) - o;
        }
        if (kind == 3) { o = s.center + (0.5f * s.radius) * vector3(S(), S(), S()) * 0.5f; d = randdir(); }             // inside
        if (kind == 4) { vector3 n = randdir().normalized(); o = s.center + s.radius * n; d = randdir(); }               // on surface
        if (kind == 5) { tmax = 0.5f + 30.0f * U(); }                                                                      // bounded tmax
        if (kind == 7) { vector3 n = randdir().normalized(); o = s.center + s.radius * n; d = -1.0f * n + 0.3f * randdir(); } // refracted-style, from surface inward
        r = ray(o, d);
        hit_info h;
        bool     ret = sphereHit(s, r, tmin, tmax, &h);
        put(s.center); put(s.radius); put(r.origin); put(r.direction); put(tmin); put(tmax);
        put(ret ? 1.0f : 0.0f);
        if (ret) { put(h.distance); put(h.point); put(h.normal); } else { for (int k = 0; k < 7; ++k) put(0.0f); }
        w.row();
    }
}

static void kat_optics(const std::string& dir)
{
    Writer w(dir + "/kat_optics.f32");
    for (int i = 0; i < 2000; ++i) {
        vector3 v = randdir() * (0.2f + 3.0f * U());
        vector3 n = randdir().normalized();
        float   ni = (i & 1) ? 1.5f : 1.0f / 1.5f;
        if (i % 7 == 0) ni = 0.5f + 2.0f * U();
        if (i % 5 == 0) { // near the critical angle / total internal reflection
            vector3 t = n.cross(randdir()).normalized();
            float   c = 0.745f + 0.002f * S();
            v         = (-c) * n + sqrtf(1.0f - c * c) * t;
            ni        = 1.5f;
        }
        float   cosine = U();
        float   ior    = (i % 3 == 0) ? 1.0f + 1.5f * U() : 1.5f;
        vector3 refl   = _reflect(v, n);
        vector3 refr(0, 0, 0);
        bool    ok = _refract(v, n, ni, &refr);
        if (!ok) refr = vector3(0, 0, 0);
        float sch = _schlick(cosine, ior);
        put(v); put(n); put(ni); put(cosine); put(ior);
        put(refl); put(ok ? 1.0f : 0.0f); put(refr); put(sch);
        w.row();
    }
}

static void kat_background(const std::string& dir)
{
    Writer w(dir + "/kat_background.f32");
    for (int i = 0; i < 1000; ++i) {
        vector3 d = randdir() * (0.01f + 20.0f * U());
        ray     r(vector3(S(), S(), S()), d);
        put(d); put(_background(r));
        w.row();
    }
}

// Injected-draw lists are regenerated from (row, stream) on both sides (here and in oracle/pyoracle.py kat_draws)
// so that the fixtures only have to store the row number and the mode.
static std::vector<float> make_draws(uint32_t row, uint32_t stream, int n, int mode)
{
    std::vector<float> d(n);
    for (int i = 0; i < n; ++i) {
        uint32_t ctr[4] = { (uint32_t)i / 4u, row, stream, 0x4B4154u }, key[2] = { 0u, 0u }, r[4];
        ro_philox4x32_10(ctr, key, r);
        d[i] = ro_u32_to_unit(r[i & 3]);
    }
    if (mode == 1) for (int i = 0; i < n; ++i) d[i] = 0.97f + 0.0299f * d[i];                   // every try rejected: 21-try exhaustion
    if (mode == 2) for (int i = 0; i < n && i < 30; ++i) d[i] = (i % 3 == 0) ? 0.999f : 0.001f; // ten rejections, then random
    return d;
}

static void kat_samplers(const std::string& dir)
{
    Writer w(dir + "/kat_samplers.f32");
    for (int i = 0; i < 1500; ++i) {
        int                mode = i % 50 == 0 ? 1 : (i % 50 == 1 ? 2 : 0);
        std::vector<float> d    = make_draws(i, 1, 64, mode);
        ro_rng g;
        ro_rng_inject(&g, d.data(), 64);
        refinject::set_rng(&g);
        vector3  ps = randomInUnitSphere();
        uint32_t us = g.inject_pos;
        ro_rng_inject(&g, d.data(), 64);
        vector3  pd = randomOnUnitDisk();
        uint32_t ud = g.inject_pos;
        refinject::set_rng(nullptr);
        put((float)i); put((float)mode);
        put(ps); put((float)us); put(pd); put((float)ud);
        w.row();
    }
}

static void kat_camera(const std::string& dir)
{
    Writer wc(dir + "/kat_camera.f32");
    Writer wr(dir + "/kat_get_ray.f32");
    for (int i = 0; i < 200; ++i) {
        float   vfov = 10.0f + 80.0f * U(), aspect = 0.5f + 2.0f * U(), aperture = 0.5f * U(), focus = 1.0f + 20.0f * U();
        vector3 pos(15 * S(), 15 * S(), 15 * S()), up(0, 1, 0), lookat(3 * S(), 3 * S(), 3 * S());
        if (i == 0) { vfov = 20; aspect = 1200.0f / 800.0f; aperture = 0.1f; focus = 10.0f; pos = vector3(13, 2, 3); lookat = vector3(0, 0, 0); }
        if (i == 1) { vfov = 20; aspect = 3840.0f / 2160.0f; aperture = 0.1f; focus = 10.0f; pos = vector3(13, 2, 3); lookat = vector3(0, 0, 0); }
        if (i == 2) { vfov = 20; aspect = 2000.0f / 1000.0f; aperture = 0.1f; focus = 10.0f; pos = vector3(13, 2, 3); lookat = vector3(0, 0, 0); }
        if (i % 10 == 9) up = randdir();
        Camera c(vfov, aspect, aperture, focus, pos, up, lookat);
        put(vfov); put(aspect); put(aperture); put(focus); put(pos); put(up); put(lookat);
        put(c.origin); put(c.leftCorner); put(c.horizontal); put(c.vertical); put(c.u); put(c.v); put(c.w); put(c.lensRadius);
        wc.row();
        for (int k = 0; k < 6; ++k) {
            int                mode = (k == 3 && i % 20 == 0) ? 1 : 0;
            uint32_t           rowid = (uint32_t)(i * 6 + k);
            std::vector<float> d    = make_draws(rowid, 2, 42, mode);
            float s = U(), t = U();
            ro_rng g;
            ro_rng_inject(&g, d.data(), 42);
            refinject::set_rng(&g);
            ray r = c.getRay(s, t);
            refinject::set_rng(nullptr);
            put(c.origin); put(c.leftCorner); put(c.horizontal); put(c.vertical); put(c.u); put(c.v); put(c.w); put(c.lensRadius);
            put(s); put(t); put((float)rowid); put((float)mode);
            put(r.origin); put(r.direction); put((float)g.inject_pos);
            wr.row();
        }
    }
}

static void kat_scatter(const std::string& dir)
{
    Writer w(dir + "/kat_scatter.f32");
    for (int i = 0; i < 1500; ++i) {
        material_t m;
        int        kind = i % 3;
        m.type          = kind == 0 ? MATERIAL_DIFFUSE : (kind == 1 ? MATERIAL_METAL : MATERIAL_GLASS);
        m.albedo        = vector3(U(), U(), U());
        m.blur          = (i % 11 == 0) ? 0.0f : 0.5f * U();
        m.refractionIndex = (i % 13 == 0) ? 1.0f + U() : 1.5f;
        vector3 n = randdir().normalized();
        vector3 c(10 * S(), 10 * S(), 10 * S());
        float   radius = 0.2f + U();
        hit_info h;
        h.point    = c + radius * n;
        h.normal   = (h.point - c) / radius;
        h.distance = 0.01f + 20.0f * U();
        h.material = m;
        vector3 d = randdir() * (0.05f + 4.0f * U());
        if (i % 4 != 3 && d.dot(h.normal) > 0) d = -d;       // mostly arriving from outside; one in four from inside
        if (kind == 2 && i % 5 == 0) {                         // grazing / near-critical for glass from inside
            vector3 t = h.normal.cross(randdir()).normalized();
            float   cs = 0.745f + 0.003f * S();
            d = cs * h.normal + sqrtf(1.0f - cs * cs) * t;
            d = d * (0.5f + U());
        }
        ray r(h.point - h.distance * d, d);
        int                mode = i % 97 == 0 ? 1 : (i % 97 == 1 ? 2 : 0);
        std::vector<float> dr   = make_draws(i, 3, 64, mode);
        ro_rng g;
        ro_rng_inject(&g, dr.data(), 64);
        refinject::set_rng(&g);
        vector3 att(0, 0, 0);
        ray     sc(vector3(0, 0, 0), vector3(0, 0, 0));
        bool    ret = materialScatter(m, r, h, &att, &sc);
        refinject::set_rng(nullptr);
        put((float)m.type); put(m.albedo); put(m.blur); put(m.refractionIndex);
        put(r.origin); put(r.direction);
        put(h.distance); put(h.point); put(h.normal);
        put((float)i); put((float)mode);
        put(ret ? 1.0f : 0.0f); put(att); put(sc.origin); put(sc.direction); put((float)g.inject_pos);
        w.row();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One frame through the reference's own functions with the shared Philox stream injected.
//   .primary.u32 : sphere index of the first _sceneHit of every (sample, pixel), via the reference's sphereHit in
//                  _sceneHit's loop order (raytracer.cpp:291-300; the reference itself returns no index)
//   .accum.f32   : per-pixel colour sums, the loop of raytracer.cpp:175-186 around Camera::getRay and _color
//   .fb.u32      : the framebuffer written by the reference's own _renderJob (one tile covering the image)
static int cmd_frame(int argc, char** argv)
{
    if (argc < 11) return 2;
    std::vector<sphere_t> scene = load_scene(argv[2]);
    std::string prefix = argv[3];
    unsigned cols = atoi(argv[4]), rows = atoi(argv[5]), spp_begin = atoi(argv[6]), spp = atoi(argv[7]), depth = atoi(argv[8]);
    uint64_t seed = strtoull(argv[9], nullptr, 10);
    bool     recursive = atoi(argv[10]) != 0;
    Camera   cam = book_camera(cols, rows);
    const sphere_t* sc = scene.data();
    uint32_t n = (uint32_t)scene.size();

    std::vector<uint32_t> primary((size_t)spp * rows * cols);
    std::vector<float>    accum((size_t)rows * cols * 3);
    refinject::Schedule   sched;
    sched.seed = seed;
    uint64_t rays_miss = 0, total_events = 0;

    for (unsigned y = 0; y < rows; ++y) {
        for (unsigned x = 0; x < cols; ++x) {
            vector3  color(0, 0, 0);
            uint32_t pixel = y * cols + x;
            for (unsigned s = 0; s < spp; ++s) {
                ro_rng g;
                ro_rng_begin_sample(&g, seed, pixel, spp_begin + s);
                refinject::set_rng(&g);
                float u = float(x + pk::random()) / float(cols);
                float v = float(y + pk::random()) / float(rows);
                ray   r = cam.getRay(u, v);

                // primary-hit index: _sceneHit's loop, reference sphereHit
                uint32_t idx = 0xFFFFFFFFu;
                float    closest = (std::numeric_limits<float>::max)();
                for (uint32_t i = 0; i < n; ++i) {
                    hit_info tmp;
                    if (sphereHit(sc[i], r, 0.001f, closest, &tmp)) { closest = tmp.distance; idx = i; }
                }
                hit_info chk;
                bool     hit = _sceneHit(sc, n, r, 0.001f, (std::numeric_limits<float>::max)(), &chk);
                if (hit != (idx != 0xFFFFFFFFu) || (hit && chk.distance != closest)) { fprintf(stderr, "index loop disagrees with _sceneHit\n"); return 1; }
                if (!hit) rays_miss++;
                primary[(size_t)s * rows * cols + pixel] = idx;

                if (recursive) color += _color_recursive(r, sc, n, 0, depth);
                else           color += _color(r, sc, n, 0, depth);
                sched.entries.push_back({ pixel, spp_begin + s, g.event });
                total_events += g.event;
            }
            accum[3 * (size_t)pixel + 0] = color.x;
            accum[3 * (size_t)pixel + 1] = color.y;
            accum[3 * (size_t)pixel + 2] = color.z;
        }
    }
    refinject::set_rng(nullptr);
    sched.entries.push_back({ 0, 0, 0 }); // sentinel: the walk looks one entry ahead

    // The reference's own per-tile job over the whole image (needs spp_begin == 0: _renderJob counts from 0).
    std::vector<uint32_t> fb((size_t)rows * cols, 0xCCCCCCCCu);
    if (spp_begin == 0) {
        std::atomic<uint32_t> blockCount{ 0 };
        RenderThreadContext   ctx;
        ctx.camera = &cam; ctx.scene = sc; ctx.sceneSize = n; ctx.framebuffer = fb.data(); ctx.rows = rows; ctx.cols = cols;
        ctx.num_aa_samples = spp; ctx.max_ray_depth = depth; ctx.blockID = 0; ctx.blockSize = (cols > rows ? cols : rows);
        ctx.xOffset = 0; ctx.yOffset = 0; ctx.blockCount = &blockCount; ctx.totalBlocks = 1; ctx.debug = false; ctx.recursive = recursive;
        refinject::set_schedule(&sched);
        _renderJob(&ctx, 0);
        refinject::set_schedule(nullptr);
    }

    FILE* f;
    f = fopen((prefix + ".primary.u32").c_str(), "wb"); fwrite(primary.data(), 4, primary.size(), f); fclose(f);
    f = fopen((prefix + ".accum.f32").c_str(), "wb");   fwrite(accum.data(), 4, accum.size(), f);     fclose(f);
    f = fopen((prefix + ".fb.u32").c_str(), "wb");      fwrite(fb.data(), 4, fb.size(), f);           fclose(f);
    printf("{\"n\": %u, \"cols\": %u, \"rows\": %u, \"spp\": %u, \"primary_miss\": %llu, \"scatter_events\": %llu}\n",
           n, cols, rows, spp, (unsigned long long)rays_miss, (unsigned long long)total_events);
    return 0;
}

// Unmodified renderScene (thread pool, tiles, mt19937 draws): statistical reference image + reference timing.
static int cmd_render_ref(int argc, char** argv)
{
    if (argc < 12) return 2;
    std::vector<sphere_t> flat = load_scene(argv[2]);
    unsigned cols = atoi(argv[4]), rows = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]), threads = atoi(argv[8]), block = atoi(argv[9]);
    bool     recursive = atoi(argv[10]) != 0;
    refinject::set_mt_seed((uint32_t)strtoul(argv[11], nullptr, 10));
    Scene scene;
    for (const sphere_t& s : flat) scene.objects.push_back(new Sphere(s.center, s.radius, new material_t(s.material)));
    Camera cam = book_camera(cols, rows);
    std::vector<uint32_t> fb((size_t)rows * cols, 0xCCCCCCCCu);
    auto t0 = std::chrono::steady_clock::now();
    renderScene(scene, cam, rows, cols, fb.data(), spp, depth, threads, block, false, recursive);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    FILE* f = fopen(argv[3], "wb"); fwrite(fb.data(), 4, fb.size(), f); fclose(f);
    printf("{\"seconds\": %.6f, \"samples\": %llu, \"threads\": %u}\n", sec, (unsigned long long)rows * cols * spp, threads);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_harness sizes|kat|frame|render_ref ...\n"); return 2; }
    std::string cmd = argv[1];
    if (cmd == "sizes") {
        printf("{\"sphere_t\": %zu, \"material_t\": %zu, \"hit_info\": %zu, \"Camera\": %zu, \"RenderThreadContext\": %zu}\n",
               sizeof(sphere_t), sizeof(material_t), sizeof(hit_info), sizeof(Camera), sizeof(RenderThreadContext));
        return 0;
    }
    if (cmd == "kat" && argc >= 3) {
        std::string dir = argv[2];
        kat_sphere_hit(dir); kat_optics(dir); kat_background(dir); kat_samplers(dir); kat_camera(dir); kat_scatter(dir);
        return 0;
    }
    if (cmd == "frame") return cmd_frame(argc, argv);
    if (cmd == "render_ref") return cmd_render_ref(argc, argv);
    fprintf(stderr, "unknown command\n");
    return 2;
}
