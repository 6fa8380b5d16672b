// rtx_plan.h -- host-only planning helpers shared by rtx_host.cpp (no HIP) and rtx_api.hip.
#pragma once
#include <stdint.h>
#include <vector>

namespace rtx
{
constexpr uint32_t kMaxPlanChunks = 63;                // chunks of an automatic plan (TraceParams::chunk_tab holds their 64 boundaries)
// Fills tab[0 .. n] with the chunk boundaries of a launch (relative to spp_begin) and returns n; see rtx_host.cpp.

// Uniform grid over the sphere table (RTX_FILTER_GRID): see rtx_host.cpp, build_grid.
struct GridDesc {
    float    gmin[3];        // lower corner of the grid box
    float    inv_h[3];       // cells per world unit
    float    h[3];           // cell size
    uint32_t dim[3];         // cells per axis (0, 0, 0: no grid -- every sphere is on the big list)
    float    pad;            // positional slack the cells were registered with (world units); the traversal stops early only
                             // when the closest hit lies more than this before the cell's exit
    uint32_t n_big;          // spheres too large for the grid: tested for every ray
};
struct GridHost {
    GridDesc              desc;
    std::vector<uint32_t> cell_start;   // dim.x * dim.y * dim.z + 1 offsets into items
    std::vector<uint32_t> items;        // sphere indices, cell by cell (ascending inside a cell)
    std::vector<uint32_t> big;          // indices of the big spheres (ascending)
};
void build_grid(uint32_t n, const float* cx, const float* cy, const float* cz, const float* radius, double per_cell, GridHost* out);

uint32_t plan_chunks(uint32_t spp_count, uint64_t npix, uint64_t lanes, uint32_t items_per_lane, uint32_t max_chunks, uint32_t tail_factor, uint64_t max_items, uint32_t* tab);
}
