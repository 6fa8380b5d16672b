// Host code of librtx under AddressSanitizer + UBSan (CPU only): the grid builder, the chunk planner, the scene generator and the
// camera on random and degenerate inputs.  Built and run by tests/test_host_sanitizers.py from raytracing_amd/csrc/rtx_host.cpp.
#include "../../include/rtx.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <vector>

static int fails = 0;
#define REQUIRE(c) do { if (!(c)) { printf("FAILED %s (line %d)\n", #c, __LINE__); ++fails; } } while (0)

int main()
{
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    // ---- rtx_build_grid ----------------------------------------------------------------------------------
    for (int iter = 0; iter < 160; ++iter) {
        const uint32_t n = iter < 6 ? (uint32_t)iter : (uint32_t)(U(rng) * U(rng) * 3000);
        std::vector<float> cx(n), cy(n), cz(n), r(n);
        const double spread = std::pow(10.0, U(rng) * 8 - 3), thin = U(rng) < 0.5 ? 0.01 : 1.0;
        for (uint32_t i = 0; i < n; ++i) {
            cx[i] = (float)((U(rng) - 0.5) * spread); cy[i] = (float)((U(rng) - 0.5) * spread * thin); cz[i] = (float)((U(rng) - 0.5) * spread);
            r[i]  = (float)(spread * std::pow(10.0, -U(rng) * 4) * 0.05);
            const double q = U(rng);
            if (q < 0.01) r[i] = 0.0f; else if (q < 0.02) r[i] = -r[i]; else if (q < 0.03) r[i] = (float)spread * 50.0f;
            else if (q < 0.035) cx[i] = std::numeric_limits<float>::quiet_NaN(); else if (q < 0.04) cy[i] = std::numeric_limits<float>::infinity();
            else if (q < 0.06 && i) { cx[i] = cx[i - 1]; cy[i] = cy[i - 1]; cz[i] = cz[i - 1]; r[i] = r[i - 1]; }
        }
        const uint32_t density = 1u + (uint32_t)(U(rng) * U(rng) * 800);
        rtx_grid_info info;
        REQUIRE(rtx_build_grid(n, cx.data(), cy.data(), cz.data(), r.data(), density, &info, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0) == RTX_OK);
        const uint64_t ncell = (uint64_t)info.dim[0] * info.dim[1] * info.dim[2];
        REQUIRE(info.dim[0] <= 256 && info.dim[1] <= 256 && info.dim[2] <= 256 && info.n_big <= n);
        std::vector<uint32_t> start(ncell + 1), items(info.n_items + 1), big(info.n_big + 1), tend(3 * ncell + 1);
        REQUIRE(rtx_build_grid(n, cx.data(), cy.data(), cz.data(), r.data(), density, &info, start.data(), start.size(), items.data(), items.size(),
                               big.data(), big.size(), tend.data(), tend.size()) == RTX_OK);
        if (ncell) REQUIRE(start[0] == 0 && start[ncell] == info.n_items);
        for (uint64_t c = 0; c < ncell; ++c)
            REQUIRE(start[c] <= tend[3 * c] && tend[3 * c] <= tend[3 * c + 1] && tend[3 * c + 1] <= tend[3 * c + 2] && tend[3 * c + 2] == start[c + 1]);
        for (uint64_t k = 0; k < info.n_items; ++k) REQUIRE(items[k] < n);
        std::vector<char> seen(n, 0);
        for (uint32_t k = 0; k < info.n_big; ++k) { REQUIRE(big[k] < n); if (big[k] < n) seen[big[k]] = 1; }
        for (uint64_t k = 0; k < info.n_items; ++k) if (items[k] < n) seen[items[k]] = 1;
        for (uint32_t i = 0; i < n; ++i) REQUIRE(seen[i]);                       // every sphere is reachable: on the big list or in a cell
        // too-small arrays are refused, not overrun
        if (info.n_items > 1) REQUIRE(rtx_build_grid(n, cx.data(), cy.data(), cz.data(), r.data(), density, &info, nullptr, 0, items.data(), info.n_items - 1, nullptr, 0, nullptr, 0) != RTX_OK);
    }
    // ---- rtx_plan_chunks ---------------------------------------------------------------------------------
    for (int iter = 0; iter < 4000; ++iter) {
        const uint32_t spp = 1u + (uint32_t)(std::pow(U(rng), 3) * 10000);
        const uint64_t npix = 1u + (uint64_t)(std::pow(U(rng), 2) * 9000000), lanes = 64u << (int)(U(rng) * 13);
        const uint32_t ipl = 1u + (uint32_t)(U(rng) * 4096), maxc = 1u + (uint32_t)(U(rng) * 80), tf = 1u + (uint32_t)(U(rng) * 100);
        const uint64_t max_items = npix * (1u + (uint64_t)(U(rng) * 70));
        uint32_t tab[64] = { 0 }, nch = 0;
        REQUIRE(rtx_plan_chunks(spp, npix, lanes, ipl, maxc, tf, max_items, tab, 64, &nch) == RTX_OK);
        REQUIRE(nch >= 1 && nch <= 63 && nch <= maxc && tab[0] == 0 && tab[nch] == spp && (uint64_t)nch * npix <= max_items);
        for (uint32_t k = 0; k < nch; ++k) REQUIRE(tab[k + 1] > tab[k]);
    }
    // ---- rtx_random_scene / rtx_camera_init --------------------------------------------------------------
    for (int gh : { 0, 1, 11, 40 }) {
        const uint32_t cap = rtx_scene_capacity(gh);
        std::vector<float> f[9]; for (auto& v : f) v.resize(cap);
        std::vector<uint32_t> t(cap);
        uint32_t n = 0;
        REQUIRE(rtx_random_scene(7, gh, 0.8f, 0.95f, &n, f[0].data(), f[1].data(), f[2].data(), f[3].data(), t.data(), f[4].data(), f[5].data(), f[6].data(), f[7].data(), f[8].data()) == RTX_OK);
        REQUIRE(n <= cap && n >= 4);
    }
    { uint32_t n = 0; float x[8]; uint32_t t[8]; REQUIRE(rtx_random_scene(7, -1, 0.8f, 0.95f, &n, x, x, x, x, t, x, x, x, x, x) != RTX_OK); }     // refused, nothing written
    rtx_camera cam;
    const float pos[3] = { 13, 2, 3 }, up[3] = { 0, 1, 0 }, look[3] = { 0, 0, 0 };
    REQUIRE(rtx_camera_init(&cam, 20.0f, 1.5f, 0.1f, 10.0f, pos, up, look) == RTX_OK && cam.lens_radius == 0.05f);
    printf(fails ? "host fuzz: %d failures\n" : "host fuzz ok\n", fails);
    return fails ? 1 : 0;
}
