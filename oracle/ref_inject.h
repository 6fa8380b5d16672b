// Random-source selection for the compiled reference (see ref_utils_host.cpp).  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "rt_oracle.h"

#include <cstddef>
#include <cstdint>
#include <vector>

namespace refinject
{
struct ScheduleEntry { uint32_t pixel, sample, events; };
struct Schedule { uint64_t seed; std::vector<ScheduleEntry> entries; };

void set_rng(ro_rng* g);               // this thread draws from g (nullptr: back to mt19937)
void set_schedule(const Schedule* s);  // this thread walks s, one entry per sample, in call order
void set_mt_seed(uint32_t seed);
float draw();                          // what `_random( _gen )` of utils.cu:70 evaluates to in this build
} // namespace refinject
