/* oracle_tsan.c -- the oracle's tile pool (oracle/rt_oracle.c render_frame / render_worker: the role of the reference's
 * thread_pool.cpp:344-382 workers taking _renderJob tiles) under ThreadSanitizer: several frames on 1 .. 8 threads, with and
 * without a primary-hit map and an explicit chunk plan; every multi-threaded frame must equal the one-thread frame bit for bit.
 * SURVEY s5 ("race detection"): the reference's pool shares an unsynchronised mt19937 and unlocks its SpinLockGuard temporaries
 * at once (utils.cu:26-28,70; thread_pool.cpp:123); the restatement draws from a counter-based stream and must be race-free. */
#include "../../oracle/rt_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(void)
{
    const uint32_t cap = ro_random_scene_capacity(11);
    ro_sphere* scene = (ro_sphere*)malloc(sizeof(ro_sphere) * cap);
    const uint32_t n = ro_random_scene(1, 11, 0.8f, 0.95f, scene);
    const uint32_t cols = 96, rows = 64, spp = 3, depth = 50;
    ro_camera cam;
    ro_vec3 pos = { 13, 2, 3 }, up = { 0, 1, 0 }, at = { 0, 0, 0 };
    ro_camera_init(&cam, 20.0f, (float)cols / (float)rows, 0.1f, 10.0f, pos, up, at);
    float*    ref  = (float*)calloc((size_t)rows * cols * 3, sizeof(float));
    float*    got  = (float*)calloc((size_t)rows * cols * 3, sizeof(float));
    uint32_t* pref = (uint32_t*)calloc((size_t)rows * cols * spp, sizeof(uint32_t));
    uint32_t* pgot = (uint32_t*)calloc((size_t)rows * cols * spp, sizeof(uint32_t));
    ro_stats  st0, st;
    if (ro_render(scene, n, &cam, rows, cols, 0, spp, 0, depth, 5, 0, 1, 16, ref, pref, &st0) != 0) return 2;
    const uint32_t starts[3] = { 0, 2, 3 };
    int bad = 0;
    for (uint32_t threads = 2; threads <= 8; threads += 3) {
        for (int plan = 0; plan < 2; ++plan) {
            memset(got, 0, sizeof(float) * rows * cols * 3);
            int rc = plan ? ro_render_plan(scene, n, &cam, rows, cols, 0, starts, 2, depth, 5, 0, threads, 16, got, pgot, &st)
                          : ro_render(scene, n, &cam, rows, cols, 0, spp, 0, depth, 5, 0, threads, 16, got, pgot, &st);
            if (rc != 0) return 2;
            if (!plan && memcmp(ref, got, sizeof(float) * rows * cols * 3) != 0) bad++;          /* (the plan sums in another order) */
            if (memcmp(pref, pgot, sizeof(uint32_t) * rows * cols * spp) != 0) bad++;
            if (st.rays != st0.rays || st.samples != st0.samples) bad++;
        }
    }
    free(scene); free(ref); free(got); free(pref); free(pgot);
    if (bad) { printf("oracle tsan: %d mismatches\n", bad); return 1; }
    printf("oracle tsan ok\n");
    return 0;
}
