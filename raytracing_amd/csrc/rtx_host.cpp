// rtx_host.cpp -- host-only entry points of include/rtx.h: camera derivation, the seeded book-1 scene generator
// and the PPM writer.  None of them touches a GPU.  Compiled with contraction off: the camera must come out
// bit-identical to what the reference's Camera constructor (camera.h:20-41) computes with the same arguments.
#include "../../include/rtx.h"
#include "rtx_plan.h"

#include <algorithm>

#include <cmath>
#include <cstdio>
#include <vector>
#include <cstring>

namespace
{
struct H3 { float x, y, z; };
inline H3    h3(float x, float y, float z) { return H3{ x, y, z }; }
inline H3    operator-(H3 a, H3 b) { return h3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline H3    operator*(float t, H3 v) { return h3(t * v.x, t * v.y, t * v.z); }
inline H3    cross(H3 a, H3 b) { return h3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }  // vector_cuda.h:61-66
inline H3    unit(H3 v) { float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); return h3(v.x / l, v.y / l, v.z / l); } // vector_cuda.h:53-59
inline void  st3(float* d, H3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

// Philox4x32-10, the generator shared with the kernels (trace_kernel.hip) -- here it drives the scene layout.
void philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct SceneDraws {
    uint32_t key[2];
    uint32_t n = 0;
    float    next()
    {
        uint32_t ctr[4] = { n >> 2, 0u, 0u, 0x5CE9Eu }, r[4];
        philox(ctr, key, r);
        return (float)(r[n++ & 3u] >> 8) * (1.0f / 16777216.0f);
    }
};
} // namespace

// Sample chunking.  A work item is (chunk, pixel): a lane adds the chunk's samples of its pixel in registers, in sample
// order, and stores one partial sum; reduce_kernel adds a pixel's partial sums in chunk order -- deterministic, no atomics.
// One chunk reproduces the reference's summation order exactly (raytracer.cpp:184); more chunks trade that for a short
// tail: when the work counter runs dry every lane still finishes its current item while more and more lanes idle.
// Round 1 used uniform chunks (8 samples on the bench frame: 64 chunks, 983 MB of partial sums).  What the tail needs is
// small items at the END only, so the plan is geometric: chunk sizes shrink by r per chunk down to s_min, which is sized
// for `items_per_lane` items per lane.  r follows from the requirement that an item of chunk k that takes h mean item
// durations (pixels on glass run up to 50 bounces per sample) still ends before the launch does: the work behind chunk k
// must cover h item durations of every lane, i.e. sum_{j>k} B_j >= h rho B_k with rho = lanes / pixels, so
// r / (1 - r) = h rho.  tab[0..nch] receives the boundaries relative to spp_begin; at most max_items work items.
uint32_t rtx::plan_chunks(uint32_t spp_count, uint64_t npix, uint64_t lanes, uint32_t items_per_lane, uint32_t max_chunks, uint32_t tail_factor, uint64_t max_items, uint32_t* tab)
{
    uint64_t allowed = std::min<uint64_t>((uint64_t)std::max<uint32_t>(max_chunks, 1u), rtx::kMaxPlanChunks);
    allowed = std::max<uint64_t>(1, std::min<uint64_t>(allowed, std::min<uint64_t>(spp_count, max_items / npix)));
    const double rho = (double)lanes / (double)npix;
    const double hr  = (double)tail_factor * rho;
    const double r   = std::min(0.96, std::max(0.2, hr / (1.0 + hr)));
    // one-sample items pay the item overhead (counter, fetch, partial-sum store) once per sample: measured worth it only in
    // launches of a few dozen samples (1200x800: 24 spp 5.8 vs 5.6 Gsamples/s, but 50 spp 5.8 vs 6.6, 200 spp 7.4 vs 7.8)
    const uint64_t min_chunk = spp_count >= 48u ? 2u : 1u;
    uint64_t s_min = std::max<uint64_t>(min_chunk, ((uint64_t)spp_count * npix + lanes * (uint64_t)items_per_lane - 1) / (lanes * (uint64_t)items_per_lane));
    for (;;) {
        uint32_t nch = 0, done = 0;
        double   b = std::max<double>((double)s_min, std::floor((double)spp_count * (1.0 - r) + 0.5));
        tab[0] = 0;
        while (done < spp_count && nch < allowed) {
            uint32_t len = (uint32_t)std::max<double>((double)s_min, std::floor(b));
            if (len > spp_count - done || nch + 1 == allowed) len = spp_count - done;
            done += len;
            tab[++nch] = done;
            b *= r;
        }
        // the last allowed chunk swallowed the rest: acceptable only if it is not larger than its predecessor
        const bool ok = nch < 2 || (tab[nch] - tab[nch - 1]) <= (tab[nch - 1] - tab[nch - 2]) || allowed <= 2;
        if (ok) return nch;
        s_min = s_min + (s_min + 1) / 2;          // fewer, larger small chunks
        if (s_min >= spp_count) {                 // degenerate: an even split
            for (uint32_t k = 0; k <= allowed; ++k) tab[k] = (uint32_t)((uint64_t)spp_count * k / allowed);
            return (uint32_t)allowed;
        }
    }
}

// Uniform grid for RTX_FILTER_GRID (an acceleration structure over the SoA table; no counterpart in the
// reference, whose _sceneHit is the linear loop of raytracer.cpp:285-304).  The grid only decides WHICH spheres a ray is
// tested against; the test itself stays the exact sphere.cpp:68-102 arithmetic and the winner the minimum of (t, index), so
// results cannot change as long as no sphere that the REFERENCE'S fp32 test accepts is skipped.  That test is blurred: with
// u = 2^-24, L = |o - c| and the uncontracted evaluation order of sphere.cpp:72-78,
//     |disc_fp32 - a (r^2 - dist^2)|  <=  20 u a (L^2 + r^2)        (dist: distance of the centre from the ray's line)
// (oc: u per component; b = oc.d: 4 u L |d|; b^2: 9 u a L^2; oc.oc: 5 u L^2; c: 6 u L^2 + 2 u r^2; a c: u a (10 L^2 + 6 r^2); the
// last subtraction: u a (L^2 + r^2)), so a sphere can be "hit" by rays that pass it at up to
//     blur(r, L) = sqrt(r^2 + 2^-19 (L^2 + r^2)) - r                (2^-19 = 32 u: the bound with a margin)
// beyond its surface: 0.0007 for r = 0.2 at L = 15, 0.03 at L = 100, 0.9 at L = 800 -- far away the reference renders noise,
// and the grid has to reproduce it.  The accepted root t then lies on the part of the ray inside the ball of radius
// r + blur(r, L), hence inside the sphere's bounding box grown by that much.  Construction:
//   * three registration TIERS for origin-to-sphere distances up to L_k = L/4, L/2, L with L = two diagonals of the binned
//     spheres' bounds (less in large scenes, see below):
//     in tier k a sphere is registered in every cell that its bounding box grown by pad + blur(r, L_k) overlaps.  A cell's list
//     is stored tier by tier (tier 0, then what tier 1 adds, then what tier 2 adds).  A ray reads the lists 0..k of its cells
//     when the farthest point its walk can reach -- the box's exit or a hit already found on the big list -- is at most
//     tier_s[k] = L_k - (cell diagonal + largest binned radius + pad + largest blur of tier k) from its origin.  An origin from
//     which a binned sphere can be farther than L_2 (distance to the box + the box's diagonal > L_2: far-away cameras) is beyond
//     the registrations: such a ray is clipped against the box grown by the blur of that distance and, if it enters, scans the
//     whole table;
//   * `pad` = 2^-12 of the largest coordinate or extent covers the walk's own arithmetic: the slab clip and the accumulated
//     cell-exit parameters are off by at most 256 roundings of a parameter whose distance equivalent is <= 2 diagonals
//     (far_l), i.e. < 2^-14 of the extent, and b / a by 4 u L; a hit point that the walk places in the neighbouring cell is
//     still found there, and the walk stops early only when the closest hit so far lies more than 2 pad before the exit of the
//     cell just finished (an untested sphere's registered stretch of the ray starts after that exit, and its root lies on
//     that stretch);
//   * the grid box is the binned spheres' bounds grown by the largest registration radius, so that every registered stretch
//     of a ray lies inside the box; an axis that is one cell thick stays one cell thick;
//   * spheres whose tier-0 box would cover more than an eighth of the cells (at least 8, at most 512: the ground sphere) are
//     not binned but tested for every ray; the grid is then laid out again over the rest.
// Resolution: `per_cell` spheres per cell (tuning key "grid_density_pct", default 0.35: measured flat between 0.125 and 0.5 on
// the book-1 and 20 164-sphere scenes, profiles/r02c_grid_filter.txt), cells as cubic as the extent allows, at most 256 cells per axis.
double rtx::grid_blur(double r, double l)
{
    r = std::fabs(r);
    return std::sqrt(r * r + std::ldexp(l * l + r * r, -19)) - r;
}

void rtx::build_grid(uint32_t n, const float* cx, const float* cy, const float* cz, const float* radius, double per_cell, GridHost* out)
{
    GridHost& G = *out;
    G = GridHost();
    memset(&G.desc, 0, sizeof(G.desc));
    for (int t = 0; t < 3; ++t) { G.tier_l[t] = 0.0; G.tier_grow[t] = 0.0; }
    std::vector<uint32_t> cand;
    std::vector<uint8_t>  is_big(n, 0);
    uint32_t n_wide = 0;                          // spheres put on the big list for being wide, not huge
    for (int round = 0; round < 8; ++round) {
        // bounds of the spheres that are (still) meant for the grid
        double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 }, r_max = 0.0;
        cand.clear();
        for (uint32_t i = 0; i < n; ++i) {
            const double c[3] = { cx[i], cy[i], cz[i] }, r = std::fabs((double)radius[i]);
            if (!(std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]) && std::isfinite(r))) { is_big[i] = 1; continue; }   // the exact test decides
            if (is_big[i]) continue;
            cand.push_back(i);
            r_max = std::max(r_max, r);
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], c[k] - r); hi[k] = std::max(hi[k], c[k] + r); }
        }
        if (cand.empty()) break;
        double ext[3], big_coord = 0.0;
        for (int k = 0; k < 3; ++k) { ext[k] = hi[k] - lo[k]; big_coord = std::max(big_coord, std::max(std::max(std::fabs(lo[k]), std::fabs(hi[k])), ext[k])); }
        const double pad = std::max(big_coord * std::ldexp(1.0, -12), 1e-30);
        for (int k = 0; k < 3; ++k) { lo[k] -= 2.0 * pad; hi[k] += 2.0 * pad; ext[k] = hi[k] - lo[k]; }
        const double diag = std::sqrt(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);
        // cell size: (volume * per_cell / count)^(1/3), but a thin extent is one cell thick and then no longer counts as a dimension
        double h = std::cbrt(ext[0] * ext[1] * ext[2] * per_cell / (double)cand.size());
        for (int pass = 0; pass < 3; ++pass) {
            double vol = per_cell / (double)cand.size();
            int    dims = 0;
            for (int k = 0; k < 3; ++k) if (ext[k] > h) { vol *= ext[k]; dims++; }
            if (dims == 0) break;
            h = std::pow(vol, 1.0 / dims);
        }
        // Tier limits: the outermost reaches two diagonals, less where the blur of the smallest sphere would exceed a quarter of a
        // cell by then (large scenes: the box and every far list grow with it), but never less than 1.1 diagonals, so that a ray
        // that starts inside the box is always served; the inner tiers halve it.
        double r_min = r_max;
        for (uint32_t i : cand) r_min = std::min(r_min, std::fabs((double)radius[i]));
        double l_out = 2.0 * diag;
        if (grid_blur(r_min, l_out) > 0.25 * h) {
            double lo_l = 0.0, hi_l = l_out;                // blur is monotonic in l: bisect
            for (int it = 0; it < 60; ++it) { const double mid = 0.5 * (lo_l + hi_l); (grid_blur(r_min, mid) > 0.25 * h ? hi_l : lo_l) = mid; }
            l_out = std::max(1.1 * diag, lo_l);
        }
        const double tier_l[3] = { 0.25 * l_out, 0.5 * l_out, l_out };
        double grow[3] = { pad, pad, pad };
        for (uint32_t i : cand)
            for (int t = 0; t < 3; ++t) grow[t] = std::max(grow[t], pad + grid_blur(radius[i], tier_l[t]));
        uint32_t dim[3];
        double   hk[3];
        for (int k = 0; k < 3; ++k) {
            dim[k] = (uint32_t)std::min(256.0, std::max(1.0, std::floor(ext[k] / std::max(h, 1e-300))));
            lo[k] -= grow[2]; hi[k] += grow[2];             // every registered stretch of a ray lies inside the box
            hk[k]  = (hi[k] - lo[k]) / dim[k];
        }
        // which spheres are too large for this grid?
        const uint64_t all_cells = (uint64_t)dim[0] * dim[1] * dim[2];
        const uint64_t kBigCells = std::max<uint64_t>(8, std::min<uint64_t>(512, all_cells / 8));
        bool changed = false;
        for (uint32_t i : cand) {
            const double r = std::fabs((double)radius[i]) + pad + grid_blur(radius[i], tier_l[0]);
            uint64_t cells = 1;
            for (int k = 0; k < 3; ++k) cells *= (uint64_t)std::min<double>(dim[k], std::floor(2.0 * r / hk[k]) + 2.0);
            if (cells > kBigCells) { is_big[i] = 1; changed = true; }
        }
        // A few spheres much larger than the rest also go on the big list: there a sphere costs one test per ray with all lanes
        // busy (about one wave-instruction per ray); in the grid it is tested in every one of its cells that a ray crosses, at the
        // walk's lane utilisation -- and it can multiply the grid's thin extent (book-1's three unit spheres over a field of
        // r = 0.2 make a second layer of cells that rays cross without meeting anything: 82 against 50 ms per 500 spp).  Outliers
        // are radii above 2.5 medians, the 16 largest at most.
        if (!changed && n_wide < 16u) {
            std::vector<uint32_t> order(cand);
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return std::fabs(radius[x]) > std::fabs(radius[y]); });
            const double median = std::fabs((double)radius[order[order.size() / 2]]);
            for (uint32_t i : order) {
                if (n_wide >= 16u || !(std::fabs((double)radius[i]) > 2.5 * median)) break;
                is_big[i] = 1; changed = true; ++n_wide;
            }
        }
        if (changed) continue;                    // their removal may shrink the bounds a lot (the ground sphere): again
        GridDesc& D = G.desc;
        for (int k = 0; k < 3; ++k) { D.gmin[k] = (float)lo[k]; D.h[k] = (float)hk[k]; D.inv_h[k] = (float)(1.0 / hk[k]); D.dim[k] = dim[k]; }
        D.pad = (float)pad;
        const double cell_diag = std::sqrt(hk[0] * hk[0] + hk[1] * hk[1] + hk[2] * hk[2]);
        for (int t = 0; t < 3; ++t) { G.tier_l[t] = tier_l[t]; G.tier_grow[t] = grow[t]; }
        // rounded down a little: the kernel compares fp32 distances with these
        for (int t = 0; t < 2; ++t) D.tier_s[t] = (float)std::max(0.0, (tier_l[t] - (cell_diag + r_max + grow[t])) * (1.0 - 1e-5));
        D.far_l    = (float)(tier_l[2] * (1.0 - 1e-5));
        D.box_diag = (float)(std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2])) * (1.0 + 1e-5));
        D.r_min    = (float)(r_min * (1.0 - 1e-5));
        const size_t ncells = (size_t)dim[0] * dim[1] * dim[2];
        auto range = [&](uint32_t i, int k, int t, uint32_t* a, uint32_t* b) {
            const double c = k == 0 ? cx[i] : (k == 1 ? cy[i] : cz[i]), r = std::fabs((double)radius[i]) + pad + grid_blur(radius[i], tier_l[t]);
            const double ga = (c - r - lo[k]) / hk[k], gb = (c + r - lo[k]) / hk[k];
            *a = (uint32_t)std::min<double>(dim[k] - 1, std::max(0.0, std::floor(ga)));
            *b = (uint32_t)std::min<double>(dim[k] - 1, std::max(0.0, std::floor(gb)));
        };
        std::vector<uint32_t> count(ncells * 3, 0u), cursor;
        for (int fill = 0; fill < 2; ++fill) {
            if (fill) {
                G.cell_start.assign(ncells + 1, 0u);
                G.tier_end.assign(ncells * 3, 0u);
                cursor.assign(ncells * 3, 0u);
                uint32_t run = 0;
                for (size_t c = 0; c < ncells; ++c) {
                    G.cell_start[c] = run;
                    for (int t = 0; t < 3; ++t) { cursor[c * 3 + t] = run; run += count[c * 3 + t]; G.tier_end[c * 3 + t] = run; }
                }
                G.cell_start[ncells] = run;
                G.items.assign(run, 0u);
            }
            for (uint32_t i : cand) {
                uint32_t a[3][3], b[3][3];
                for (int t = 0; t < 3; ++t) for (int k = 0; k < 3; ++k) range(i, k, t, &a[t][k], &b[t][k]);
                for (uint32_t z = a[2][2]; z <= b[2][2]; ++z)
                    for (uint32_t y = a[2][1]; y <= b[2][1]; ++y)
                        for (uint32_t x = a[2][0]; x <= b[2][0]; ++x) {
                            const uint32_t p[3] = { x, y, z };
                            int tier = 2;
                            for (int t = 1; t >= 0; --t) {
                                bool in = true;
                                for (int k = 0; k < 3; ++k) in = in && p[k] >= a[t][k] && p[k] <= b[t][k];
                                if (in) tier = t;
                            }
                            const size_t c = ((size_t)z * dim[1] + y) * dim[0] + x;
                            if (fill) G.items[cursor[c * 3 + tier]++] = i; else count[c * 3 + tier]++;
                        }
            }
        }
        break;
    }
    for (uint32_t i = 0; i < n; ++i) if (is_big[i] || G.desc.dim[0] == 0u) G.big.push_back(i);
    G.desc.n_big = (uint32_t)G.big.size();
}

extern "C" int rtx_build_grid(uint32_t n, const float* cx, const float* cy, const float* cz, const float* radius, uint32_t density_pct,
                              rtx_grid_info* info, uint32_t* cell_start, uint64_t cell_start_cap, uint32_t* items, uint64_t items_cap,
                              uint32_t* big, uint64_t big_cap, uint32_t* tier_end, uint64_t tier_end_cap)
{
    if (!info || (n && !(cx && cy && cz && radius)) || density_pct < 1 || density_pct > 6400) return RTX_ERR_ARG;
    rtx::GridHost G;
    rtx::build_grid(n, cx, cy, cz, radius, density_pct / 100.0, &G);
    for (int k = 0; k < 3; ++k) { info->gmin[k] = G.desc.gmin[k]; info->inv_h[k] = G.desc.inv_h[k]; info->h[k] = G.desc.h[k]; info->dim[k] = G.desc.dim[k]; }
    info->pad = G.desc.pad; info->n_big = G.desc.n_big; info->n_items = G.items.size();
    info->box_diag = G.desc.box_diag; info->r_min = G.desc.r_min;
    for (int t = 0; t < 3; ++t) { info->tier_s[t] = t < 2 ? G.desc.tier_s[t] : G.desc.far_l; info->tier_l[t] = G.tier_l[t]; info->tier_grow[t] = G.tier_grow[t]; }
    if (cell_start) { if (cell_start_cap < G.cell_start.size()) return RTX_ERR_ARG; std::copy(G.cell_start.begin(), G.cell_start.end(), cell_start); }
    if (items)      { if (items_cap < G.items.size()) return RTX_ERR_ARG; std::copy(G.items.begin(), G.items.end(), items); }
    if (big)        { if (big_cap < G.big.size()) return RTX_ERR_ARG; std::copy(G.big.begin(), G.big.end(), big); }
    if (tier_end)   { if (tier_end_cap < G.tier_end.size()) return RTX_ERR_ARG; std::copy(G.tier_end.begin(), G.tier_end.end(), tier_end); }
    return RTX_OK;
}

extern "C" int rtx_plan_chunks(uint32_t spp_count, uint64_t n_pixels, uint64_t lanes, uint32_t items_per_lane, uint32_t max_chunks,
                               uint32_t tail_factor, uint64_t max_items, uint32_t* starts, uint32_t cap, uint32_t* n_chunks)
{
    if (!starts || !n_chunks || cap < rtx::kMaxPlanChunks + 1u || spp_count == 0 || n_pixels == 0 || lanes == 0 || items_per_lane == 0 || tail_factor == 0) return RTX_ERR_ARG;
    if (n_pixels > max_items) return RTX_ERR_ARG;
    *n_chunks = rtx::plan_chunks(spp_count, n_pixels, lanes, items_per_lane, max_chunks, tail_factor, max_items, starts);
    return RTX_OK;
}

extern "C" int rtx_camera_init(rtx_camera* cam, float vfov, float aspect, float aperture, float focus, const float pos[3], const float up[3], const float lookat[3])
{
    if (!cam || !pos || !up || !lookat) return RTX_ERR_ARG;
    const float pi = 3.14159265358979323846f;                 // utils.h:70, a float literal
    cam->lens_radius = aperture / 2.0f;                       // camera.h:28
    float theta      = vfov * pi / 180.0f;                    // camera.h:30
    float halfHeight = tanf(theta / 2);                       // camera.h:31 writes "(float)tan( theta / 2 )" with a float argument: in C++ with <cmath> in scope that is the float
                                                              // overload, i.e. tanf -- which is what the reference's own object code computes: the camera KAT made by the compiled
                                                              // reference (tests/golden, test_camera_matches_reference_constructor) fails with the double tan (tried in r3)
    float halfWidth  = aspect * halfHeight;
    H3 origin = h3(pos[0], pos[1], pos[2]);
    H3 w = unit(origin - h3(lookat[0], lookat[1], lookat[2])); // camera.h:35-37
    H3 u = unit(cross(h3(up[0], up[1], up[2]), w));
    H3 v = cross(w, u);
    H3 lc = origin - (halfWidth * focus) * u - (halfHeight * focus) * v - focus * w;   // camera.h:39
    st3(cam->origin, origin);
    st3(cam->left_corner, lc);
    st3(cam->horizontal, (2 * halfWidth * focus) * u);        // camera.h:40
    st3(cam->vertical, (2 * halfHeight * focus) * v);         // camera.h:41
    st3(cam->u, u); st3(cam->v, v); st3(cam->w, w);
    return RTX_OK;
}

extern "C" uint32_t rtx_scene_capacity(int grid_half) { return grid_half < 0 ? 4u : (uint32_t)(4 * grid_half * grid_half + 4); }

extern "C" int rtx_random_scene(uint64_t seed, int grid_half, float p_diffuse, float p_metal, uint32_t* n_out,
                                float* cx, float* cy, float* cz, float* radius, uint32_t* mat_type,
                                float* ar, float* ag, float* ab, float* blur, float* ior)
{
    if (!n_out || !cx || !cy || !cz || !radius || !mat_type || !ar || !ag || !ab || !blur || !ior || grid_half < 0) return RTX_ERR_ARG;
    SceneDraws g;
    g.key[0] = (uint32_t)seed; g.key[1] = (uint32_t)(seed >> 32);
    uint32_t n = 0;
    auto push = [&](float x, float y, float z, float r, uint32_t type, float a0, float a1, float a2, float bl, float io) {
        cx[n] = x; cy[n] = y; cz[n] = z; radius[n] = r; mat_type[n] = type; ar[n] = a0; ag[n] = a1; ab[n] = a2; blur[n] = bl; ior[n] = io; n++;
    };
    push(0, -1000.0f, 0, 1000, RTX_MATERIAL_DIFFUSE, 0.5f, 0.5f, 0.5f, 1.0f, 1.0f);              // RayTracing.cpp:186
    for (int a = -grid_half; a < grid_half; a++) {
        for (int b = -grid_half; b < grid_half; b++) {                                            // :188-203
            float material = g.next();
            float x = a + 0.9f * g.next();
            float z = b + 0.9f * g.next();
            H3    dlt = h3(x, 0.2f, z) - h3(4.0f, 0.2f, 0.0f);
            if (sqrtf(dlt.x * dlt.x + dlt.y * dlt.y + dlt.z * dlt.z) > 0.9f) {
                if (material < p_diffuse) {
                    float r0 = g.next(), r1 = g.next(), g0 = g.next(), g1 = g.next(), b0 = g.next(), b1 = g.next();
                    push(x, 0.2f, z, 0.2f, RTX_MATERIAL_DIFFUSE, r0 * r1, g0 * g1, b0 * b1, 1.0f, 1.0f);
                } else if (material > p_metal) {
                    float r0 = g.next(), g0 = g.next(), b0 = g.next(), bl = g.next();
                    push(x, 0.2f, z, 0.2f, RTX_MATERIAL_METAL, 0.5f * (1 + r0), 0.5f * (1 + g0), 0.5f * (1 + b0), 0.5f * bl, 1.0f);
                } else {
                    push(x, 0.2f, z, 0.2f, RTX_MATERIAL_GLASS, 1, 1, 1, 1.0f, 1.5f);
                }
            }
        }
    }
    push(-4, 1, 0, 1.0f, RTX_MATERIAL_DIFFUSE, 0.4f, 0.2f, 0.1f, 1.0f, 1.0f);                      // :205-207
    push(0, 1, 0, 1.0f, RTX_MATERIAL_GLASS, 0.4f, 0.2f, 0.1f, 1.0f, 1.5f);
    push(4, 1, 0, 1.0f, RTX_MATERIAL_METAL, 0.7f, 0.6f, 0.5f, 0.0f, 1.0f);
    *n_out = n;
    return RTX_OK;
}

extern "C" int rtx_write_ppm(const char* path, uint32_t cols, uint32_t rows, const uint32_t* fb)
{
    if (!path || !fb) return RTX_ERR_ARG;
    FILE* f = fopen(path, "w");
    if (!f) return RTX_ERR_ARG;
    fprintf(f, "P3\n");
    fprintf(f, "%d %d\n", (int)cols, (int)rows);
    fprintf(f, "255\n");
    for (uint32_t y = 0; y < rows; y++) {
        for (uint32_t x = 0; x < cols; x++) {
            uint32_t rgb = fb[(size_t)y * cols + x];
            fprintf(f, "%d %d %d\n", (int)((rgb >> 24) & 0xFF), (int)((rgb >> 16) & 0xFF), (int)((rgb >> 8) & 0xFF));
        }
    }
    fflush(f);
    fclose(f);
    return RTX_OK;
}

extern "C" int rtx_write_ppm_p6(const char* path, uint32_t cols, uint32_t rows, const uint32_t* fb)
{
    if (!path || !fb) return RTX_ERR_ARG;
    FILE* f = fopen(path, "wb");
    if (!f) return RTX_ERR_ARG;
    fprintf(f, "P6\n%d %d\n255\n", (int)cols, (int)rows);
    std::vector<unsigned char> row((size_t)cols * 3);
    for (uint32_t y = 0; y < rows; y++) {
        for (uint32_t x = 0; x < cols; x++) {
            const uint32_t rgb = fb[(size_t)y * cols + x];
            row[3 * (size_t)x + 0] = (unsigned char)(rgb >> 24);
            row[3 * (size_t)x + 1] = (unsigned char)(rgb >> 16);
            row[3 * (size_t)x + 2] = (unsigned char)(rgb >> 8);
        }
        if (fwrite(row.data(), 1, row.size(), f) != row.size()) { fclose(f); return RTX_ERR_ARG; }
    }
    fclose(f);
    return RTX_OK;
}
