"""Randomised parity: small random scenes (overlapping spheres, tiny and huge radii, all three materials, cameras inside
spheres, zero aperture) through the HIP backend and the oracle must agree bit for bit, with all three candidate filters
(VALU, MFMA, uniform grid) and with zero filter misses.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import raytracing_amd
    raytracing_amd.lib()
    return raytracing_amd


def _random_table(rng, n):
    t = np.zeros((n, 10), dtype=np.float32)
    spread = rng.choice([2.0, 10.0, 60.0])
    t[:, 0:3] = rng.uniform(-spread, spread, size=(n, 3))
    t[:, 3] = np.exp(rng.uniform(np.log(0.02), np.log(4.0), size=n))                # radii over two decades
    if n > 1 and rng.random() < 0.5:
        t[0, 0:4] = (0.0, -500.0 - spread, 0.0, 500.0)                            # a ground sphere
    if n > 2 and rng.random() < 0.5:
        t[1, 0:4] = t[2, 0:4]                                                     # two identical spheres: the lower index must win
    t[:, 4] = rng.integers(1, 4, size=n)                                            # material type
    t[:, 5:8] = rng.uniform(0.0, 1.0, size=(n, 3))
    t[:, 8] = rng.uniform(0.0, 1.0, size=n) * (rng.random(n) < 0.7)                 # blur, often 0
    t[:, 9] = rng.uniform(1.05, 2.4, size=n)
    return t


@pytest.mark.parametrize("case", range(48))
def test_random_scene_matches_oracle(ra, oracle, case):
    rng = np.random.default_rng(1000 + case)
    n = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 64, 127, 129, 200]))
    table = _random_table(rng, n)
    cols, rows, spp, depth = int(rng.integers(17, 90)), int(rng.integers(9, 60)), int(rng.integers(1, 5)), int(rng.choice([1, 4, 50]))
    if case % 4 == 3:                                      # camera at the centre of a sphere (inside it)
        pos = table[n // 2, 0:3].copy()
    else:
        pos = rng.uniform(-1.5, 1.5, size=3).astype(np.float32) * np.float32(table[:, 0:3].max() + 3.0)
    look = table[0, 0:3] + np.float32(0.25)
    aperture = np.float32(0.0 if case % 3 == 0 else rng.uniform(0.01, 0.5))
    args = (np.float32(rng.uniform(15, 80)), np.float32(cols) / np.float32(rows), aperture, np.float32(rng.uniform(1, 30)), pos.astype(np.float32), (0, 1, 0), look.astype(np.float32))
    cam, ocam = ra.Camera(*args), oracle.camera(*args)
    rec = bool(case % 5 == 4)
    o_accum, o_primary, o_st = oracle.render(table, ocam, rows, cols, 0, spp, depth, 77 + case, chunk_spp=spp, recursive=rec, want_primary=True)
    for kind in (ra.FILTER_MFMA, ra.FILTER_VALU, ra.FILTER_GRID):
        ctx = ra.Context(0)
        try:
            ctx.set_filter(kind)
            ctx.upload_scene(table)
            ctx.set_camera(cam)
            flags = ra.FLAG_VALIDATE_FILTER | (ra.FLAG_RECURSIVE if rec else 0)
            primary, st = ctx.render(cols, rows, 0, spp, depth, 77 + case, flags=flags, chunk_spp=spp, want_primary=True)
            accum = ctx.read_accum()
        finally:
            ctx.close()
        assert st["filter_misses"] == 0, (case, kind)
        assert (primary == o_primary).all(), (case, kind)
        same = (bits(accum) == bits(o_accum)) | (np.isnan(accum) & np.isnan(o_accum))
        assert same.all(), (case, kind, int((~same).sum()))
        for k in ("samples", "rays", "term_miss", "term_absorb", "term_depth"):
            assert st[k] == o_st[k], (case, kind, k)


_ORDERED_SEEN = []


@pytest.mark.parametrize("case", range(16))
def test_random_scene_second_launch_matches_oracle_whatever_the_work_order(ra, oracle, case):
    """The second and third launches of a frame run a cost-ordered work list when the frame carries enough work (round 3).
    Random scenes, cameras, sizes, sample ranges, depths, -r, the three candidate searches, odd chunk limits: each launch must
    equal the oracle summing in the chunks the library reports, bit for bit -- the order in which the work is handed out must
    never show."""
    rng = np.random.default_rng(5000 + case)
    n = int(rng.choice([3, 31, 64, 129, 300]))
    table = _random_table(rng, n)
    cols, rows = int(rng.integers(280, 460)), int(rng.integers(190, 300))       # enough work per resident lane for three-sample chunks
    spp, depth, seed = int(rng.integers(12, 30)), int(rng.choice([3, 8, 8, 50])), 900 + case
    pos = rng.uniform(-1.5, 1.5, size=3).astype(np.float32) * np.float32(table[:, 0:3].max() + 3.0)
    look = table[0, 0:3] + np.float32(0.25)
    args = (np.float32(rng.uniform(15, 80)), np.float32(cols) / np.float32(rows), np.float32(rng.uniform(0.0, 0.3)), np.float32(rng.uniform(1, 30)), pos, (0, 1, 0), look.astype(np.float32))
    cam, ocam = ra.Camera(*args), oracle.camera(*args)
    rec = bool(case % 6 == 5)
    kind = (ra.FILTER_MFMA, ra.FILTER_GRID, ra.FILTER_VALU, ra.FILTER_AUTO)[case % 4]
    ctx = ra.Context(0)
    try:
        ctx.set_filter(kind)
        ctx.set_tuning("order_share_pct", int(rng.choice([200, 400])))
        ctx.set_tuning("order_chunks", int(rng.choice([2, 5, 16, 63])))
        ctx.upload_scene(table)
        ctx.set_camera(cam)
        flags = ra.FLAG_RECURSIVE if rec else 0
        begin = 0
        for launch in range(3):
            count = spp if launch < 2 else spp + 3
            primary, st = ctx.render(cols, rows, begin, count, depth, seed, flags=flags, want_primary=True)
            _ORDERED_SEEN.append(ctx.last_launch_ordered())
            assert not (launch == 0 and _ORDERED_SEEN[-1])
            chunk_spp, starts = ctx.last_chunk_plan()
            assert chunk_spp == 0 and starts[-1] == count
            o_accum, o_primary, o_st = oracle.render(table, ocam, rows, cols, begin, count, depth, seed, recursive=rec, want_primary=True, chunk_starts=starts)
            accum = ctx.read_accum()
            assert (primary == o_primary).all(), (case, launch)
            same = (bits(accum) == bits(o_accum)) | (np.isnan(accum) & np.isnan(o_accum))
            assert same.all(), (case, launch, int((~same).sum()))
            for k in ("samples", "rays", "term_miss", "term_absorb", "term_depth"):
                assert st[k] == o_st[k], (case, launch, k)
            if launch == 1:
                begin = spp                                  # the third launch: the next sample range of the same frame
    finally:
        ctx.close()
    if case == 15:
        assert sum(_ORDERED_SEEN) >= 10, sum(_ORDERED_SEEN)      # the cases do exercise ordered launches
