// The random source behind the reference's own utils.cu (compiled by ref_utils_wrap.cpp).  TEST INFRASTRUCTURE ONLY:
// linked into oracle/_ref/ref_harness, never into the product.
//
// pk::random / randomInUnitSphere / randomOnUnitDisk ARE utils.cu:65-94; only the uniform floats behind
// `_random( _gen )` come from one of three sources selected per thread by the harness:
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

// ---- the source behind `_random( _gen )` of utils.cu:70 (hook in ref_utils_wrap.cpp) -------------------------------
//
// The shared stream is structured (rt_oracle.h): a sampler try is one Philox block, x, y, z in words 0, 1, 2.  The
// reference writes `vector3( random(), random(), random() )` (utils.cu:79,90), whose evaluation order C++ leaves to the
// compiler; g++ evaluates the LAST argument first.  draw() therefore hands out a try's words in the order in which this
// compiler asks for them (probed once, below), so that the reference's point has the components the stream defines --
// the loop, its exit condition and the 21-try limit are the reference's own object code.
namespace refinject
{
enum SamplerMode { kBare = 0, kSphere = 1, kDisk = 2 };
thread_local int      t_mode = kBare;
thread_local uint32_t t_calls = 0, t_base = 0;

static int  s_probe_next = 0;
static int  probe_next() { return s_probe_next++; }
struct ProbeTriple { int a, b, c; ProbeTriple(int x, int y, int z) : a(x), b(y), c(z) {} };
static bool last_argument_first()
{
    static const bool v = [] { s_probe_next = 0; ProbeTriple p(probe_next(), probe_next(), probe_next()); return p.c == 0; }();
    return v;
}

static ro_rng* current_rng() { return t_rng ? t_rng : (t_schedule ? &t_sched_rng : nullptr); }

float draw()
{
    ro_rng* g = current_rng();
    if (t_mode == kBare || !g) {
        if (t_rng) return ro_random(t_rng);
        if (t_schedule) return ro_random(scheduled_rng_for_bare_call());
        static thread_local std::uniform_real_distribution<float> dist(0.0f, 1.0f);
        return dist(thread_mt());
    }
    const uint32_t width = t_mode == kSphere ? 3u : 2u;
    const uint32_t c = t_calls++, t = c / width, k = c % width;
    const uint32_t comp = last_argument_first() ? (width - 1u - k) : k;
    g->draws++;
    if (g->inject) {                                     // function KATs: a flat list, a try's components in x, y, z order
        const uint32_t i = t_base + t * width + comp;
        return i < g->inject_n ? g->inject[i] : 0.5f;
    }
    float b[4];
    if (t_mode == kSphere) {                             // event e, try t -> block t, words 0..2
        ro_rng_block(g, g->event, t, b);
        return b[comp];
    }
    if (t == 0) { ro_rng_block(g, 0, 0, b); return b[2 + comp]; }                    // disk try 0: words 2, 3 of block 0
    ro_rng_block(g, 0, (t + 1) / 2, b);
    return b[((t & 1) ? 0 : 2) + comp];
}

struct SamplerScope {
    SamplerScope(int mode)
    {
        ro_rng* g = current_rng();
        t_mode = mode; t_calls = 0;
        if (g) { t_base = g->inject_pos; if (mode == kSphere && !g->inject) g->event++; }
    }
    ~SamplerScope()
    {
        ro_rng* g = current_rng();
        if (g && g->inject) g->inject_pos = t_base + t_calls;
        t_mode = kBare;
    }
};
} // namespace refinject

namespace pk
{
vector3 ref_randomInUnitSphere();      // utils.cu:74-83, compiled in ref_utils_wrap.cpp
vector3 ref_randomOnUnitDisk();        // utils.cu:85-94

vector3 randomInUnitSphere()
{
    refinject::SamplerScope scope(refinject::kSphere);
    return ref_randomInUnitSphere();
}

vector3 randomOnUnitDisk()
{
    refinject::SamplerScope scope(refinject::kDisk);
    return ref_randomOnUnitDisk();
}
} // namespace pk
