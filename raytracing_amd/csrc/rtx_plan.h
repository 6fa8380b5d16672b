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
    float    pad;            // positional slack of the walk (world units): every registration is grown by at least this, and the
                             // traversal stops early only when the closest hit lies more than twice this before the cell's exit
    uint32_t n_big;          // spheres too large for the grid: tested for every ray
    float    tier_s[2];      // a ray that cannot get farther than tier_s[k] from its origin inside the box reads the lists 0..k of its cells
    float    far_l;          // the outermost tier's limit: an origin from which some binned sphere may be farther than this (its distance
                             // from the box + box_diag > far_l) is too far for the registrations -- such a ray scans the whole table
    float    box_diag;       // diagonal of the grid box
    float    r_min;          // smallest binned radius: bounds the blur of every binned sphere from above
};
struct GridHost {
    GridDesc              desc;
    std::vector<uint32_t> cell_start;   // dim.x * dim.y * dim.z + 1 offsets into items
    std::vector<uint32_t> tier_end;     // 3 per cell: ends of the tier-0, tier-1 and tier-2 parts of the cell's list (the last equals the next cell's start)
    std::vector<uint32_t> items;        // sphere indices, cell by cell, tier by tier (ascending inside a tier)
    std::vector<uint32_t> big;          // indices of the big spheres (ascending)
    double                tier_l[3];    // largest origin-to-sphere distance that tier k's registrations are valid for
    double                tier_grow[3]; // pad + the largest blur of tier k (world units)
};
// the registration radius of a sphere of radius r for ray origins up to l away: see build_grid
double grid_blur(double r, double l);
void build_grid(uint32_t n, const float* cx, const float* cy, const float* cz, const float* radius, double per_cell, GridHost* out);

uint32_t plan_chunks(uint32_t spp_count, uint64_t npix, uint64_t lanes, uint32_t items_per_lane, uint32_t max_chunks, uint32_t tail_factor, uint64_t max_items, uint32_t* tab);
}
