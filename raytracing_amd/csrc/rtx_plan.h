// rtx_plan.h -- host-only planning helpers shared by rtx_host.cpp (no HIP) and rtx_api.hip.
#pragma once
#include <stdint.h>

namespace rtx
{
constexpr uint32_t kMaxPlanChunks = 63;                // chunks of an automatic plan (TraceParams::chunk_tab holds their 64 boundaries)
// Fills tab[0 .. n] with the chunk boundaries of a launch (relative to spp_begin) and returns n; see rtx_host.cpp.
uint32_t plan_chunks(uint32_t spp_count, uint64_t npix, uint64_t lanes, uint32_t items_per_lane, uint32_t max_chunks, uint32_t tail_factor, uint64_t max_items, uint32_t* tab);
}
