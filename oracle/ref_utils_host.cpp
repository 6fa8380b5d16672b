// Host replacement for the reference's utils.cu, which cannot compile on Linux (utils.cu:59 typo, device-only
// symbols at :133,:147).  TEST INFRASTRUCTURE ONLY: linked into oracle/_ref/ref_harness, never into the product.
//
// pk::random / randomInUnitSphere / randomOnUnitDisk keep the semantics of utils.cu:65-94, but the uniform
// floats come from one of three sources selected per thread by the harness:
//   - a ro_rng (the shared Philox stream of oracle/rt_oracle.h), for bit-exact path comparisons;
//   - a schedule of ro_rng streams that advances by itself, for driving the reference's own _renderJob;
//   - a std::mt19937 (the reference's generator, utils.cu:26-28), for unmodified renderScene runs.
#include "ref_inject.h"

#include "utils.h"

#include <atomic>
#include <chrono>
#include <random>
#include <thread>

namespace refinject
{
thread_local ro_rng*         t_rng      = nullptr;
thread_local const Schedule* t_schedule = nullptr;
thread_local size_t          t_sched_pos = 0;
thread_local ro_rng          t_sched_rng;
static std::atomic<uint32_t> s_thread_counter{ 0 };
static uint32_t              s_mt_seed = 12345u;

void set_rng(ro_rng* g) { t_rng = g; t_schedule = nullptr; }
void set_mt_seed(uint32_t seed) { s_mt_seed = seed; }
void set_schedule(const Schedule* s)
{
    t_schedule  = s;
    t_sched_pos = 0;
    t_rng       = nullptr;
    if (s && !s->entries.empty()) {
        const ScheduleEntry& e = s->entries[0];
        ro_rng_begin_sample(&t_sched_rng, s->seed, e.pixel, e.sample);
    }
}

static std::mt19937& thread_mt()
{
    thread_local std::mt19937 gen(s_mt_seed + 7919u * s_thread_counter.fetch_add(1));
    return gen;
}

// A bare random() that arrives after a sample has consumed all of its scatter events is the first jitter draw
// of the next (pixel, sample) in _renderJob's loop order (raytracer.cpp:157-186).
static ro_rng* scheduled_rng_for_bare_call()
{
    const ScheduleEntry& e = t_schedule->entries[t_sched_pos];
    if (t_sched_rng.bare_calls == 2 && t_sched_rng.event == e.events) {
        t_sched_pos++;
        const ScheduleEntry& n = t_schedule->entries.at(t_sched_pos);
        ro_rng_begin_sample(&t_sched_rng, t_schedule->seed, n.pixel, n.sample);
    }
    return &t_sched_rng;
}
} // namespace refinject

namespace pk
{

bool delay(size_t ms)
{
    std::this_thread::sleep_for(std::chrono::milliseconds(ms));
    return true;
}

result threadSetName(const char*) { return R_NOTIMPL; }

float random()
{
    using namespace refinject;
    if (t_rng) return ro_random(t_rng);
    if (t_schedule) return ro_random(scheduled_rng_for_bare_call());
    static thread_local std::uniform_real_distribution<float> dist(0.0f, 1.0f);
    return dist(thread_mt());
}

vector3 randomInUnitSphere()
{
    using namespace refinject;
    ro_rng* g = t_rng ? t_rng : (t_schedule ? &t_sched_rng : nullptr);
    if (g) {
        ro_vec3 p = ro_random_in_unit_sphere(g);
        return vector3(p.x, p.y, p.z);
    }
    vector3      point;
    unsigned int maxTries = 20;
    do {
        float x = random(), y = random(), z = random();
        point   = 2.0f * vector3(x, y, z) - vector3(1, 1, 1);
    } while (point.squared_length() >= 1.0f && maxTries--);
    return point;
}

vector3 randomOnUnitDisk()
{
    using namespace refinject;
    ro_rng* g = t_rng ? t_rng : (t_schedule ? &t_sched_rng : nullptr);
    if (g) {
        ro_vec3 p = ro_random_on_unit_disk(g);
        return vector3(p.x, p.y, p.z);
    }
    vector3      point;
    unsigned int maxTries = 20;
    do {
        float x = random(), y = random();
        point   = 2.0f * vector3(x, y, 0.0f) - vector3(1.0f, 1.0f, 0.0f);
    } while (point.dot(point) >= 1.0f && maxTries--);
    return point;
}

} // namespace pk
