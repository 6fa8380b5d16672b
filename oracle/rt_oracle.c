/*
 * rt_oracle.c -- CPU restatement of the reference trace loop.  TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * Build with -ffp-contract=off: every product and sum below must round exactly where the reference's
 * (MSVC /fp:precise, or g++ -ffp-contract=off) does.  Each helper mirrors one operator of vector_cuda.h so the
 * association order of the reference expressions is preserved.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- vector_cuda.h operators -------------------------------------------------------------------------- */

static inline ro_vec3 v3(float x, float y, float z) { ro_vec3 r = { x, y, z }; return r; }
static inline ro_vec3 v_add(ro_vec3 a, ro_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }      /* :99-102 */
static inline ro_vec3 v_sub(ro_vec3 a, ro_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }      /* :104-107 */
static inline ro_vec3 v_mul(ro_vec3 a, ro_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }      /* :109-112, :155-161 */
static inline ro_vec3 v_scale(float t, ro_vec3 v) { return v3(t * v.x, t * v.y, t * v.z); }            /* :119-122 and :129-132 */
static inline ro_vec3 v_divs(ro_vec3 v, float t) { return v3(v.x / t, v.y / t, v.z / t); }             /* :124-127 */
static inline ro_vec3 v_neg(ro_vec3 v) { return v3(-v.x, -v.y, -v.z); }                                /* :30 */
static inline float   v_dot(ro_vec3 a, ro_vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }  /* :68-71 */
static inline float   v_sqlen(ro_vec3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }                 /* :42 */
static inline float   v_len(ro_vec3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }            /* :41 */
static inline ro_vec3 v_cross(ro_vec3 a, ro_vec3 b)                                                    /* :61-66 */
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline ro_vec3 v_normalized(ro_vec3 v)                                                          /* :45-59 */
{
    float length = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return v3(v.x / length, v.y / length, v.z / length);
}
static inline ro_vec3 ray_point(const ro_ray* r, float t) { return v_add(r->origin, v_scale(t, r->direction)); } /* ray.h:16 */

/* ---- Philox4x32-10 (Salmon et al., SC'11; Random123 constants) ------------------------------------------ */

void ro_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float ro_u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

void ro_rng_begin_sample(ro_rng* g, uint64_t seed, uint32_t pixel, uint32_t sample)
{
    memset(g, 0, sizeof(*g));
    g->key[0] = (uint32_t)seed;
    g->key[1] = (uint32_t)(seed >> 32);
    g->pixel  = pixel;
    g->sample = sample;
}

void ro_rng_inject(ro_rng* g, const float* draws, uint32_t n)
{
    memset(g, 0, sizeof(*g));
    g->inject = draws;
    g->inject_n = n;
}

static void rng_block(const ro_rng* g, uint32_t event, uint32_t block, float out[4])
{
    uint32_t ctr[4] = { event * 32u + block, g->sample, g->pixel, 0u };
    uint32_t r[4];
    ro_philox4x32_10(ctr, g->key, r);
    for (int i = 0; i < 4; ++i) out[i] = ro_u32_to_unit(r[i]);
}

void ro_rng_block(const ro_rng* g, uint32_t event, uint32_t block, float out[4]) { rng_block(g, event, block, out); }

static float rng_injected(ro_rng* g)
{
    float v = (g->inject_pos < g->inject_n) ? g->inject[g->inject_pos] : 0.5f;
    g->inject_pos++;
    g->draws++;
    return v;
}

/* utils.cu:65-72.  A bare random() is either a pixel jitter (the first two calls of a sample,
 * raytracer.cpp:177-178) or the glass reflect/refract draw (material.cu:96). */
float ro_random(ro_rng* g)
{
    if (g->inject) return rng_injected(g);
    float b[4];
    g->draws++;
    if (g->event == 0 && g->bare_calls < 2) {
        rng_block(g, 0, 0, b);
        return b[g->bare_calls++];
    }
    g->event++;
    rng_block(g, g->event, 0, b);
    return b[0];
}

/* utils.cu:74-83: <= 21 tries, the last point is returned even when it lies outside. */
ro_vec3 ro_random_in_unit_sphere(ro_rng* g)
{
    ro_vec3      point;
    unsigned int maxTries = 20;
    unsigned int t = 0;
    if (!g->inject) g->event++;
    do {
        float rx, ry, rz;
        if (g->inject) {
            rx = rng_injected(g); ry = rng_injected(g); rz = rng_injected(g);
        } else {
            float b[4];
            rng_block(g, g->event, t, b);
            rx = b[0]; ry = b[1]; rz = b[2];
            g->draws += 3;
        }
        t++;
        point = v_sub(v_scale(2.0f, v3(rx, ry, rz)), v3(1, 1, 1));
    } while (v_sqlen(point) >= 1.0f && maxTries--);
    return point;
}

/* utils.cu:85-94 */
ro_vec3 ro_random_on_unit_disk(ro_rng* g)
{
    ro_vec3      point;
    unsigned int maxTries = 20;
    unsigned int t = 0;
    do {
        float rx, ry;
        if (g->inject) {
            rx = rng_injected(g); ry = rng_injected(g);
        } else {
            float b[4];
            if (t == 0) { rng_block(g, 0, 0, b); rx = b[2]; ry = b[3]; }
            else {
                rng_block(g, 0, (t + 1) / 2, b);
                if (t & 1) { rx = b[0]; ry = b[1]; } else { rx = b[2]; ry = b[3]; }
            }
            g->draws += 2;
        }
        t++;
        point = v_sub(v_scale(2.0f, v3(rx, ry, 0.0f)), v3(1.0f, 1.0f, 0.0f));
    } while (v_dot(point, point) >= 1.0f && maxTries--);
    return point;
}

/* ---- sphere.cpp:68-102 ---------------------------------------------------------------------------------- */

int ro_sphere_hit(const ro_sphere* s, const ro_ray* r, float tmin, float tmax, ro_hit* hit)
{
    ro_vec3 oc = v_sub(r->origin, s->center);
    float   a  = v_dot(r->direction, r->direction);
    float   b  = v_dot(oc, r->direction);
    float   c  = v_dot(oc, oc) - (s->radius * s->radius);

    float discriminant = b * b - a * c;

    if (discriminant > 0) {
        float t = (-b - sqrtf(discriminant)) / a;
        if (t < tmax && t > tmin) {
            hit->distance = t;
            hit->point    = ray_point(r, t);
            hit->normal   = v_divs(v_sub(hit->point, s->center), s->radius);
            hit->material = s->material;
            return 1;
        }
        t = (-b + sqrtf(discriminant)) / a;
        if (t < tmax && t > tmin) {
            hit->distance = t;
            hit->point    = ray_point(r, t);
            hit->normal   = v_divs(v_sub(hit->point, s->center), s->radius);
            hit->material = s->material;
            return 1;
        }
    }
    return 0;
}

/* raytracer.cpp:285-304; *index = i of the last accepted hit (the reference has no index; strict t < closest
 * means the lowest index wins exact ties). */
int ro_scene_hit(const ro_sphere* scene, uint32_t n, const ro_ray* r, float tmin, float tmax, ro_hit* hit, uint32_t* index)
{
    int      rval         = 0;
    float    closestSoFar = tmax;
    uint32_t idx          = 0xFFFFFFFFu;
    ro_hit   best;
    memset(&best, 0, sizeof(best));
    for (uint32_t i = 0; i < n; ++i) {
        ro_hit tmp;
        if (ro_sphere_hit(&scene[i], r, tmin, closestSoFar, &tmp)) {
            rval         = 1;
            closestSoFar = tmp.distance;
            best         = tmp;
            idx          = i;
        }
    }
    *hit = best;
    if (index) *index = idx;
    return rval;
}

/* ---- material.cu ------------------------------------------------------------------------------------------ */

ro_vec3 ro_reflect(ro_vec3 v, ro_vec3 n)                                   /* :112-115 */
{
    return v_sub(v, v_scale(2 * v_dot(v, n), n));
}

int ro_refract(ro_vec3 v, ro_vec3 n, float ni_over_nt, ro_vec3* refracted)  /* :118-129 */
{
    ro_vec3 _v           = v_normalized(v);
    float   dt           = v_dot(_v, n);
    float   discriminant = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (discriminant > 0) {
        *refracted = v_sub(v_scale(ni_over_nt, v_sub(_v, v_scale(dt, n))), v_scale(sqrtf(discriminant), n));
        return 1;
    }
    return 0;
}

float ro_schlick(float cosine, float ior)                                   /* :132-138; pow(float,int) is double */
{
    float r0 = (1.0f - ior) / (1.0f + ior);
    r0       = r0 * r0;
    return (float)((double)r0 + (double)(1.0f - r0) * pow((double)(1.0f - cosine), 5.0));
}

static int diffuse_scatter(const ro_material* d, const ro_hit* hit, ro_vec3* attenuation, ro_ray* scattered, ro_rng* g)   /* :47-54 */
{
    ro_vec3 target       = v_add(v_add(hit->point, hit->normal), ro_random_in_unit_sphere(g));
    scattered->origin    = hit->point;
    scattered->direction = v_sub(target, hit->point);
    *attenuation         = d->albedo;
    return 1;
}

static int metal_scatter(const ro_material* m, const ro_ray* r, const ro_hit* hit, ro_vec3* attenuation, ro_ray* scattered, ro_rng* g) /* :58-65 */
{
    ro_vec3 reflected    = ro_reflect(v_normalized(r->direction), hit->normal);
    scattered->origin    = hit->point;
    scattered->direction = v_add(reflected, v_scale(m->blur, ro_random_in_unit_sphere(g)));
    *attenuation         = m->albedo;
    return v_dot(scattered->direction, hit->normal) > 0;
}

static int glass_scatter(const ro_material* gm, const ro_ray* r, const ro_hit* hit, ro_vec3* attenuation, ro_ray* scattered, ro_rng* g) /* :69-105 */
{
    ro_vec3 outwardNormal;
    ro_vec3 reflected = ro_reflect(r->direction, hit->normal);
    float   niOverNt;
    ro_vec3 refracted = v3(0, 0, 0);
    float   probability;
    float   cosine;

    *attenuation = v3(1.0f, 1.0f, 1.0f);

    if (v_dot(r->direction, hit->normal) > 0) {
        outwardNormal = v_neg(hit->normal);
        niOverNt      = gm->ior;
        cosine        = gm->ior * v_dot(r->direction, hit->normal) / v_len(r->direction);
    } else {
        outwardNormal = hit->normal;
        niOverNt      = 1.0f / gm->ior;
        cosine        = -v_dot(r->direction, hit->normal) / v_len(r->direction);
    }

    if (ro_refract(r->direction, outwardNormal, niOverNt, &refracted)) {
        probability = ro_schlick(cosine, gm->ior);
    } else {
        probability = 1.0f;
    }

    float p = ro_random(g);

    scattered->origin    = hit->point;
    scattered->direction = (p < probability) ? reflected : refracted;
    return 1;
}

int ro_material_scatter(const ro_material* m, const ro_ray* r, const ro_hit* hit, ro_vec3* attenuation, ro_ray* scattered, ro_rng* g) /* :17-40 */
{
    /* the reference passes the same object as r and scattered (raytracer.cpp:260); read r before writing */
    ro_ray in = *r;
    switch (m->type) {
        case RO_MATERIAL_DIFFUSE: return diffuse_scatter(m, hit, attenuation, scattered, g);
        case RO_MATERIAL_METAL:   return metal_scatter(m, &in, hit, attenuation, scattered, g);
        case RO_MATERIAL_GLASS:   return glass_scatter(m, &in, hit, attenuation, scattered, g);
        default:                  return -1; /* reference: printf + assert(0) */
    }
}

/* ---- raytracer.cpp:276-282 ------------------------------------------------------------------------------ */

ro_vec3 ro_background(const ro_ray* r)
{
    ro_vec3 unitDirection = v_normalized(r->direction);
    float   t             = 0.5f * (unitDirection.y + 1.0f);
    return v_add(v_scale(1.0f - t, v3(1.0f, 1.0f, 1.0f)), v_scale(t, v3(0.5f, 0.7f, 1.0f)));
}

/* ---- camera.h:20-41, :75-86 ------------------------------------------------------------------------------ */

void ro_camera_init(ro_camera* cam, float vfov, float aspect, float aperture, float focusDistance, ro_vec3 pos, ro_vec3 up, ro_vec3 lookat)
{
    const float pi = 3.14159265358979323846f;       /* utils.h:70 -- a float literal */
    cam->lensRadius = aperture / 2.0f;

    float theta      = vfov * pi / 180.0f;           /* RADIANS(), utils.h:71 */
    float halfHeight = tanf(theta / 2);   /* camera.h:31 "(float)tan( theta / 2 )": float argument -> the float overload in C++; pinned by the reference-made camera KAT */
    float halfWidth  = aspect * halfHeight;

    cam->origin = pos;
    cam->w      = v_normalized(v_sub(pos, lookat));
    cam->u      = v_normalized(v_cross(up, cam->w));
    cam->v      = v_cross(cam->w, cam->u);

    cam->leftCorner = v_sub(v_sub(v_sub(cam->origin, v_scale(halfWidth * focusDistance, cam->u)),
                                  v_scale(halfHeight * focusDistance, cam->v)),
                            v_scale(focusDistance, cam->w));
    cam->horizontal = v_scale(2 * halfWidth * focusDistance, cam->u);
    cam->vertical   = v_scale(2 * halfHeight * focusDistance, cam->v);
}

ro_ray ro_camera_get_ray(const ro_camera* cam, float s, float t, ro_rng* g)
{
    ro_vec3 rnd    = v_scale(cam->lensRadius, ro_random_on_unit_disk(g));
    ro_vec3 offset = v_add(v_scale(rnd.x, cam->u), v_scale(rnd.y, cam->v));
    ro_ray  r;
    r.origin    = v_add(cam->origin, offset);
    r.direction = v_sub(v_sub(v_add(v_add(cam->leftCorner, v_scale(s, cam->horizontal)), v_scale(1.0f - t, cam->vertical)), cam->origin), offset);
    return r;
}

/* ---- raytracer.cpp:243-273 (iterative) -------------------------------------------------------------------- */

ro_vec3 ro_color(const ro_ray* r, const ro_sphere* scene, uint32_t n, uint32_t max_depth, ro_rng* g, uint32_t* first_hit, ro_stats* st)
{
    ro_hit  hit;
    ro_vec3 attenuation;
    ro_ray  scattered = *r;
    ro_vec3 color     = v3(1, 1, 1);
    int     ended     = 0;

    if (first_hit) *first_hit = 0xFFFFFFFFu;
    for (unsigned i = 0; i < max_depth; i++) {
        uint32_t idx;
        int      h = ro_scene_hit(scene, n, &scattered, 0.001f, FLT_MAX, &hit, &idx);
        if (st) { st->rays++; st->sphere_tests += n; }
        if (i == 0 && first_hit) *first_hit = idx;
        if (h) {
            if (ro_material_scatter(&hit.material, &scattered, &hit, &attenuation, &scattered, g) > 0) {
                color = v_mul(color, attenuation);
            } else {
                if (st) st->term_absorb++;
                ended = 1;
                break;
            }
        } else {
            color = v_mul(color, ro_background(&scattered));
            if (st) st->term_miss++;
            ended = 1;
            break;
        }
    }
    if (!ended && st) st->term_depth++;
    return color;
}

/* ---- raytracer.cpp:213-240 (recursive, -r) ------------------------------------------------------------------ */

ro_vec3 ro_color_recursive(const ro_ray* r, const ro_sphere* scene, uint32_t n, uint32_t depth, uint32_t max_depth, ro_rng* g, uint32_t* first_hit, ro_stats* st)
{
    ro_hit   hit;
    uint32_t idx;
    int      h = ro_scene_hit(scene, n, r, 0.001f, FLT_MAX, &hit, &idx);
    if (st) { st->rays++; st->sphere_tests += n; }
    if (depth == 0 && first_hit) *first_hit = idx;
    if (h) {
        ro_ray  scattered;
        ro_vec3 attenuation;
        if (depth < max_depth) {
            if (ro_material_scatter(&hit.material, r, &hit, &attenuation, &scattered, g) > 0) {
                return v_mul(attenuation, ro_color_recursive(&scattered, scene, n, depth + 1, max_depth, g, NULL, st));
            }
            if (st) st->term_absorb++;
            return v3(0, 0, 0);
        }
        if (st) st->term_depth++;
        return v3(0, 0, 0);
    }
    if (st) st->term_miss++;
    return ro_background(r);
}

/* ---- raytracer.cpp:175-186 ------------------------------------------------------------------------------------ */

/* One chunk: samples [c0, c1) of the range, added in index order (color += ..., vector_cuda.h:147-153). */
static ro_vec3 chunk_sum(const ro_sphere* scene, uint32_t n, const ro_camera* cam, uint32_t rows, uint32_t cols, uint32_t x, uint32_t y,
                         uint32_t spp_begin, uint32_t c0, uint32_t c1, uint32_t max_depth, uint64_t seed, int recursive,
                         uint32_t* primary_hit, uint32_t primary_stride, ro_stats* st)
{
    ro_vec3  color = v3(0, 0, 0);
    uint32_t pixel = y * cols + x;
    for (uint32_t k = c0; k < c1; ++k) {
        ro_rng g;
        ro_rng_begin_sample(&g, seed, pixel, spp_begin + k);
        float    u = (float)(x + ro_random(&g)) / (float)cols;   /* uint + float -> float, raytracer.cpp:177 */
        float    v = (float)(y + ro_random(&g)) / (float)rows;
        ro_ray   r = ro_camera_get_ray(cam, u, v, &g);
        uint32_t first;
        ro_vec3  c = recursive ? ro_color_recursive(&r, scene, n, 0, max_depth, &g, &first, st)
                               : ro_color(&r, scene, n, max_depth, &g, &first, st);
        color = v_add(color, c);
        if (primary_hit) primary_hit[(size_t)k * primary_stride] = first;
        if (st) st->samples++;
    }
    return color;
}

ro_vec3 ro_pixel_sum(const ro_sphere* scene, uint32_t n, const ro_camera* cam, uint32_t rows, uint32_t cols,
                     uint32_t x, uint32_t y, uint32_t spp_begin, uint32_t spp_count, uint32_t chunk_spp,
                     uint32_t max_depth, uint64_t seed, int recursive, uint32_t* primary_hit, uint32_t primary_stride, ro_stats* st)
{
    ro_vec3 total = v3(0, 0, 0);
    if (chunk_spp == 0 || chunk_spp > spp_count) chunk_spp = spp_count;
    for (uint32_t c0 = 0; c0 < spp_count; c0 += chunk_spp) {
        uint32_t c1 = (c0 + chunk_spp < spp_count) ? c0 + chunk_spp : spp_count;
        total = v_add(total, chunk_sum(scene, n, cam, rows, cols, x, y, spp_begin, c0, c1, max_depth, seed, recursive, primary_hit, primary_stride, st));
    }
    return total;
}

ro_vec3 ro_pixel_sum_plan(const ro_sphere* scene, uint32_t n, const ro_camera* cam, uint32_t rows, uint32_t cols,
                          uint32_t x, uint32_t y, uint32_t spp_begin, const uint32_t* starts, uint32_t nchunks,
                          uint32_t max_depth, uint64_t seed, int recursive, uint32_t* primary_hit, uint32_t primary_stride, ro_stats* st)
{
    ro_vec3 total = v3(0, 0, 0);
    for (uint32_t k = 0; k < nchunks; ++k)
        total = v_add(total, chunk_sum(scene, n, cam, rows, cols, x, y, spp_begin, starts[k], starts[k + 1], max_depth, seed, recursive, primary_hit, primary_stride, st));
    return total;
}

/* ---- whole frame: tile jobs on a worker pool (role of raytracer.cpp:93-127 + thread_pool.cpp) ------------------ */

typedef struct {
    const ro_sphere* scene; uint32_t n; const ro_camera* cam; uint32_t rows, cols, spp_begin, spp_count, chunk_spp, max_depth;
    uint64_t seed; int recursive; uint32_t tile, tiles_x, tiles_y; float* accum; uint32_t* primary;
    const uint32_t* starts; uint32_t nchunks;          /* ro_render_plan: explicit chunk boundaries instead of chunk_spp */
    volatile uint32_t next_tile; pthread_mutex_t lock; ro_stats stats;
} render_job;

static void* render_worker(void* arg)
{
    render_job* j = (render_job*)arg;
    ro_stats    st;
    memset(&st, 0, sizeof(st));
    for (;;) {
        uint32_t t = __sync_fetch_and_add(&j->next_tile, 1u);
        if (t >= j->tiles_x * j->tiles_y) break;
        uint32_t x0 = (t % j->tiles_x) * j->tile, y0 = (t / j->tiles_x) * j->tile;
        for (uint32_t y = y0; y < y0 + j->tile && y < j->rows; ++y) {
            for (uint32_t x = x0; x < x0 + j->tile && x < j->cols; ++x) {
                size_t    p  = (size_t)y * j->cols + x;
                uint32_t* ph = j->primary ? j->primary + p : NULL;
                ro_vec3   c  = j->starts
                    ? ro_pixel_sum_plan(j->scene, j->n, j->cam, j->rows, j->cols, x, y, j->spp_begin, j->starts, j->nchunks,
                                        j->max_depth, j->seed, j->recursive, ph, j->rows * j->cols, &st)
                    : ro_pixel_sum(j->scene, j->n, j->cam, j->rows, j->cols, x, y, j->spp_begin, j->spp_count, j->chunk_spp,
                                   j->max_depth, j->seed, j->recursive, ph, j->rows * j->cols, &st);
                j->accum[3 * p + 0] = c.x;
                j->accum[3 * p + 1] = c.y;
                j->accum[3 * p + 2] = c.z;
            }
        }
    }
    pthread_mutex_lock(&j->lock);
    j->stats.samples += st.samples; j->stats.rays += st.rays; j->stats.sphere_tests += st.sphere_tests;
    j->stats.term_miss += st.term_miss; j->stats.term_absorb += st.term_absorb; j->stats.term_depth += st.term_depth;
    pthread_mutex_unlock(&j->lock);
    return NULL;
}

static int render_frame(const ro_sphere* scene, uint32_t n, const ro_camera* cam, uint32_t rows, uint32_t cols,
                        uint32_t spp_begin, uint32_t spp_count, uint32_t chunk_spp, const uint32_t* starts, uint32_t nchunks,
                        uint32_t max_depth, uint64_t seed, int recursive, uint32_t nthreads, uint32_t tile, float* accum_rgb,
                        uint32_t* primary_hit, ro_stats* st)
{
    render_job j;
    memset(&j, 0, sizeof(j));
    if (tile == 0) tile = 64;
    if (nthreads == 0) nthreads = 1;
    j.scene = scene; j.n = n; j.cam = cam; j.rows = rows; j.cols = cols; j.spp_begin = spp_begin; j.spp_count = spp_count;
    j.chunk_spp = chunk_spp; j.max_depth = max_depth; j.seed = seed; j.recursive = recursive; j.tile = tile;
    j.starts = starts; j.nchunks = nchunks;
    j.tiles_x = (cols + tile - 1) / tile; j.tiles_y = (rows + tile - 1) / tile; j.accum = accum_rgb; j.primary = primary_hit;
    pthread_mutex_init(&j.lock, NULL);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    if (!th) return -1;
    for (uint32_t i = 1; i < nthreads; ++i) pthread_create(&th[i], NULL, render_worker, &j);
    render_worker(&j);
    for (uint32_t i = 1; i < nthreads; ++i) pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&j.lock);
    if (st) *st = j.stats;
    return 0;
}

int ro_render(const ro_sphere* scene, uint32_t n, const ro_camera* cam, uint32_t rows, uint32_t cols,
              uint32_t spp_begin, uint32_t spp_count, uint32_t chunk_spp, uint32_t max_depth, uint64_t seed,
              int recursive, uint32_t nthreads, uint32_t tile, float* accum_rgb, uint32_t* primary_hit, ro_stats* st)
{
    return render_frame(scene, n, cam, rows, cols, spp_begin, spp_count, chunk_spp, NULL, 0, max_depth, seed, recursive, nthreads, tile, accum_rgb, primary_hit, st);
}

int ro_render_plan(const ro_sphere* scene, uint32_t n, const ro_camera* cam, uint32_t rows, uint32_t cols,
                   uint32_t spp_begin, const uint32_t* starts, uint32_t nchunks, uint32_t max_depth, uint64_t seed,
                   int recursive, uint32_t nthreads, uint32_t tile, float* accum_rgb, uint32_t* primary_hit, ro_stats* st)
{
    if (!starts || nchunks == 0 || starts[0] != 0) return -1;
    for (uint32_t k = 0; k < nchunks; ++k) if (starts[k + 1] < starts[k]) return -1;
    return render_frame(scene, n, cam, rows, cols, spp_begin, starts[nchunks], 0, starts, nchunks, max_depth, seed, recursive, nthreads, tile, accum_rgb, primary_hit, st);
}

/* ---- raytracer.cpp:187-197 (+ :168-171 for -d) -------------------------------------------------------------------- */

void ro_resolve(const float* accum_rgb, uint32_t rows, uint32_t cols, uint32_t spp_total, uint32_t debug_tile, uint32_t* framebuffer)
{
    float k = 1.0f / (float)spp_total;                 /* operator/=(float), vector_cuda.h:187-195 */
    for (uint32_t y = 0; y < rows; ++y) {
        for (uint32_t x = 0; x < cols; ++x) {
            size_t p = (size_t)y * cols + x;
            if (debug_tile) {
                uint32_t ty = y % debug_tile, tx = x % debug_tile;
                if (ty == 0 || ty == debug_tile - 1 || tx == 0 || tx == debug_tile - 1) { framebuffer[p] = 0xFF000000u; continue; }
            }
            float   cr = sqrtf(accum_rgb[3 * p + 0] * k);
            float   cg = sqrtf(accum_rgb[3 * p + 1] * k);
            float   cb = sqrtf(accum_rgb[3 * p + 2] * k);
            uint8_t r8 = (uint8_t)(255.99 * cr);
            uint8_t g8 = (uint8_t)(255.99 * cg);
            uint8_t b8 = (uint8_t)(255.99 * cb);
            framebuffer[p] = ((uint32_t)r8 << 24) | ((uint32_t)g8 << 16) | ((uint32_t)b8 << 8);
        }
    }
}

/* ---- RayTracing.cpp:182-213 ------------------------------------------------------------------------------------------ */

typedef struct { uint32_t key[2]; uint32_t n; } scene_rng;

static float scene_draw(scene_rng* g)
{
    uint32_t ctr[4] = { g->n >> 2, 0u, 0u, 0x5CE9Eu };
    uint32_t r[4];
    ro_philox4x32_10(ctr, g->key, r);
    return ro_u32_to_unit(r[g->n++ & 3u]);
}

static ro_material make_material(uint32_t type, ro_vec3 albedo, float blur, float ior)   /* material.h:32-36 */
{
    ro_material m; m.type = type; m.albedo = albedo; m.blur = blur; m.ior = ior; return m;
}

uint32_t ro_random_scene_capacity(int grid_half) { return (uint32_t)(4 * grid_half * grid_half + 4); }

uint32_t ro_random_scene(uint64_t seed, int grid_half, float p_diffuse, float p_metal, ro_sphere* out)
{
    scene_rng g = { { (uint32_t)seed, (uint32_t)(seed >> 32) }, 0 };
    uint32_t  n = 0;

    out[n].center = v3(0, -1000.0f, 0); out[n].radius = 1000; out[n].material = make_material(RO_MATERIAL_DIFFUSE, v3(0.5f, 0.5f, 0.5f), 1.0f, 1.0f); n++;

    for (int a = -grid_half; a < grid_half; a++) {
        for (int b = -grid_half; b < grid_half; b++) {
            float   material = scene_draw(&g);
            float   cx       = a + 0.9f * scene_draw(&g);
            float   cz       = b + 0.9f * scene_draw(&g);
            ro_vec3 center   = v3(cx, 0.2f, cz);
            if (v_len(v_sub(center, v3(4.0f, 0.2f, 0.0f))) > 0.9f) {
                if (material < p_diffuse) {
                    float r0 = scene_draw(&g), r1 = scene_draw(&g), g0 = scene_draw(&g), g1 = scene_draw(&g), b0 = scene_draw(&g), b1 = scene_draw(&g);
                    out[n].material = make_material(RO_MATERIAL_DIFFUSE, v3(r0 * r1, g0 * g1, b0 * b1), 1.0f, 1.0f);
                } else if (material > p_metal) {
                    float r0 = scene_draw(&g), g0 = scene_draw(&g), b0 = scene_draw(&g), bl = scene_draw(&g);
                    out[n].material = make_material(RO_MATERIAL_METAL, v3(0.5f * (1 + r0), 0.5f * (1 + g0), 0.5f * (1 + b0)), 0.5f * bl, 1.0f);
                } else {
                    out[n].material = make_material(RO_MATERIAL_GLASS, v3(1, 1, 1), 1.0f, 1.5f);
                }
                out[n].center = center; out[n].radius = 0.2f; n++;
            }
        }
    }

    out[n].center = v3(-4, 1, 0); out[n].radius = 1.0f; out[n].material = make_material(RO_MATERIAL_DIFFUSE, v3(0.4f, 0.2f, 0.1f), 1.0f, 1.0f); n++;
    out[n].center = v3(0, 1, 0);  out[n].radius = 1.0f; out[n].material = make_material(RO_MATERIAL_GLASS, v3(0.4f, 0.2f, 0.1f), 1.0f, 1.5f); n++;
    out[n].center = v3(4, 1, 0);  out[n].radius = 1.0f; out[n].material = make_material(RO_MATERIAL_METAL, v3(0.7f, 0.6f, 0.5f), 0.0f, 1.0f); n++;
    return n;
}
