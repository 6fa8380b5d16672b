"""The two candidate filters (VALU and MFMA) against the full scan: zero missed hits, identical images.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import raytracing_amd
    raytracing_amd.lib()
    return raytracing_amd


@pytest.mark.parametrize("scene,cols,rows,spp", [("book1", 300, 200, 4), ("glass", 200, 120, 4), ("scaled", 96, 64, 2)])
def test_filters_never_miss(ra, golden_scenes, scene, cols, rows, spp):
    """RTX_FLAG_VALIDATE_FILTER runs the reference's ascending full scan beside the filtered search for every ray."""
    results = {}
    for kind in (ra.FILTER_VALU, ra.FILTER_MFMA, ra.FILTER_GRID):
        ctx = ra.Context(0)
        ctx.set_filter(kind)
        ctx.upload_scene(golden_scenes[scene])
        ctx.set_camera(ra.book_camera(cols, rows))
        primary, st = ctx.render(cols, rows, 0, spp, 50, 77, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
        assert st["filter_misses"] == 0, (kind, st)
        if kind == ra.FILTER_GRID:
            assert st["candidates"] < 40 * st["rays"]                 # a few cells' worth of exact tests per ray, whatever the table's size
        else:
            assert st["candidates"] < 0.02 * st["sphere_tests"]       # the filter passes only a small share on to the exact test
        results[kind] = (primary, ctx.read_accum(), st)
        ctx.close()
    (p0, a0, s0), (p1, a1, s1), (p2, a2, s2) = results[ra.FILTER_VALU], results[ra.FILTER_MFMA], results[ra.FILTER_GRID]
    assert (p0 == p1).all() and (bits(a0) == bits(a1)).all() and (p0 == p2).all() and (bits(a0) == bits(a2)).all()
    for k in ("samples", "rays", "term_miss", "term_absorb", "term_depth"):
        assert s0[k] == s1[k] == s2[k]


def test_far_origins_and_large_offsets(ra, oracle, golden_scenes):
    """A scene pushed far from the coordinate origin stresses the MFMA filter's absolute margins (|C|^2 + |o|^2)."""
    table = golden_scenes["book1"].copy()
    shift = np.array([300.0, 0.0, -200.0], dtype=np.float32)
    table[:, 0:3] += shift
    cols, rows, spp = 160, 100, 2
    pos = np.array([13, 2, 3], dtype=np.float32) + shift
    cam = ra.Camera(20.0, np.float32(cols) / np.float32(rows), 0.1, 10.0, pos, (0, 1, 0), shift)
    ocam = oracle.camera(20.0, np.float32(cols) / np.float32(rows), 0.1, 10.0, pos, (0, 1, 0), shift)
    for kind in (ra.FILTER_VALU, ra.FILTER_MFMA, ra.FILTER_GRID):
        ctx = ra.Context(0)
        ctx.set_filter(kind)
        ctx.upload_scene(table)
        ctx.set_camera(cam)
        primary, st = ctx.render(cols, rows, 0, spp, 50, 5, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
        accum = ctx.read_accum()
        ctx.close()
        assert st["filter_misses"] == 0
        o_accum, o_primary, _ = oracle.render(table, ocam, rows, cols, 0, spp, 50, 5, want_primary=True)
        assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all()


@pytest.mark.parametrize("factor", [2.0 ** -10, 1024.0, 3.0e4, 1.0e7])
def test_scene_scale_sweep(ra, oracle, golden_scenes, factor):
    """The MFMA filter works on f16 operands behind a per-scene power-of-two scale: the same scene at very different
    sizes must keep zero filter misses and the oracle's bits."""
    table = golden_scenes["book1"].copy()
    f = np.float32(factor)
    table[:, 0:4] *= f                                   # centres and radii
    cols, rows, spp = 120, 80, 2
    pos, look = np.array([13, 2, 3], dtype=np.float32) * f, np.zeros(3, dtype=np.float32)
    args = (20.0, np.float32(cols) / np.float32(rows), np.float32(0.1) * f, np.float32(10.0) * f, pos, (0, 1, 0), look)
    cam, ocam = ra.Camera(*args), oracle.camera(*args)
    ctx = ra.Context(0)
    try:
        ctx.set_filter(ra.FILTER_MFMA)
        ctx.upload_scene(table)
        ctx.set_camera(cam)
        primary, st = ctx.render(cols, rows, 0, spp, 50, 5, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
        accum = ctx.read_accum()
    finally:
        ctx.close()
    assert st["filter_misses"] == 0
    assert st["candidates"] < 0.02 * st["sphere_tests"]
    o_accum, o_primary, _ = oracle.render(table, ocam, rows, cols, 0, spp, 50, 5, want_primary=True)
    # at 1e7 a few paths overflow fp32 (|oc|^2 ~ 1e34 times a ~ 1e16) in the reference's own arithmetic: both sides then
    # hold NaN in the same pixels, with the host's and the GPU's different default NaN bit patterns
    same = (bits(accum) == bits(o_accum)) | (np.isnan(accum) & np.isnan(o_accum))
    assert (primary == o_primary).all() and same.all()
    assert factor > 1.0e5 or not np.isnan(accum).any()


def test_camera_outside_f16_range_uses_valu_filter(ra, oracle, golden_scenes):
    """A camera far outside the range the scene's scale leaves for ray origins: the launch takes the VALU filter (same
    candidate count as a VALU context), results unchanged."""
    table = golden_scenes["book1"]
    cols, rows, spp = 64, 40, 1
    pos = np.array([4.0e6, 2.0e5, 1.0e6], dtype=np.float32)
    args = (0.0005, np.float32(cols) / np.float32(rows), 0.0, 10.0, pos, (0, 1, 0), (0, 0, 0))
    out = {}
    for kind in (ra.FILTER_VALU, ra.FILTER_MFMA):
        ctx = ra.Context(0)
        try:
            ctx.set_filter(kind)
            ctx.upload_scene(table)
            ctx.set_camera(ra.Camera(*args))
            primary, st = ctx.render(cols, rows, 0, spp, 50, 5, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
            out[kind] = (primary, ctx.read_accum(), st)
        finally:
            ctx.close()
    (p0, a0, s0), (p1, a1, s1) = out[ra.FILTER_VALU], out[ra.FILTER_MFMA]
    assert s0["filter_misses"] == 0 and s1["filter_misses"] == 0
    assert s0["candidates"] == s1["candidates"]
    o_accum, o_primary, _ = oracle.render(table, oracle.camera(*args), rows, cols, 0, spp, 50, 5, want_primary=True)
    assert (p1 == o_primary).all() and (bits(a1) == bits(o_accum)).all() and (p0 == p1).all()


@pytest.mark.parametrize("scene,cols,rows,spp,block", [("book1", 96, 64, 2, None), ("book1", 96, 64, 2, "256"), ("glass", 64, 40, 3, "128"), ("scaled", 64, 48, 1, None),
                                                       ("one", 40, 30, 2, None), ("empty", 40, 30, 2, None)])
def test_streamed_table_equals_the_oracle(ra, oracle, golden_scenes, monkeypatch, scene, cols, rows, spp, block):
    """The streamed-table mode (workgroup-shared LDS stage, RTX_STREAM=1 forces it for any size) against the oracle, with the
    full-scan validation on: small tables (4 chunks), the 20 164-sphere table (158 chunks), workgroups of 8, 4 and 2 waves,
    one sphere, no sphere; and again in the same process without it (the mode is chosen per launch)."""
    if scene == "one":
        table = golden_scenes["book1"][:1].copy()
    elif scene == "empty":
        table = golden_scenes["book1"][:0].copy()
    else:
        table = golden_scenes[scene]
    ctx = ra.Context(0)
    ctx.set_tuning("stream", 1)
    if block:
        ctx.set_tuning("block", int(block))
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    primary, st = ctx.render(cols, rows, 0, spp, 50, 21, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
    accum = ctx.read_accum()
    assert st["filter_misses"] == 0
    o_accum, o_primary, o_st = oracle.render(table, oracle.book_camera(cols, rows), rows, cols, 0, spp, 50, 21, want_primary=True)
    assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all()
    for k in ("samples", "rays", "term_miss", "term_absorb", "term_depth"):
        assert st[k] == o_st[k]
    # the production kernel (no validation scan), automatic chunking, -r
    _, st2 = ctx.render(cols, rows, 0, spp, 50, 21, flags=ra.FLAG_RECURSIVE)
    o2, _, _ = oracle.render(table, oracle.book_camera(cols, rows), rows, cols, 0, spp, 50, 21, recursive=True)
    np.testing.assert_allclose(ctx.read_accum(), o2, rtol=1e-5, atol=1e-6)      # chunked sums round differently
    ctx.set_tuning("stream", 0)
    ctx.render(cols, rows, 0, spp, 50, 21, chunk_spp=spp)
    assert (bits(ctx.read_accum()) == bits(accum)).all()
    ctx.close()


def _grazing_view(centre, radius, eye, cols, rows, span_ulps=2.0):
    """Camera arguments for a zero-aperture view from `eye` along a tangent of the sphere, so narrow that neighbouring
    pixels are about `span_ulps` float ulps of direction apart: the silhouette crosses the frame and thousands of primary
    rays have a discriminant b*b - a*c within a few ulps of a(|oc|^2 + r^2) of zero (sphere.cpp:77-79)."""
    c, e = np.asarray(centre, dtype=np.float64), np.asarray(eye, dtype=np.float64)
    to_c = c - e
    dist = np.linalg.norm(to_c)
    d0 = to_c / dist
    n = np.cross(d0, [0.0, 1.0, 0.0])
    if np.linalg.norm(n) < 1e-3:
        n = np.cross(d0, [1.0, 0.0, 0.0])
    n /= np.linalg.norm(n)
    sin_a = radius / dist
    tangent = np.sqrt(1.0 - sin_a * sin_a) * d0 + sin_a * n
    length = np.sqrt(dist * dist - radius * radius)
    look = e + tangent * length                                   # the tangent point
    vfov = np.degrees(rows * span_ulps * 2.0 ** -24)              # rows pixels of span_ulps * 2^-24 rad each
    return (np.float32(vfov), np.float32(cols) / np.float32(rows), 0.0, np.float32(length), e.astype(np.float32), (0, 1, 0), look.astype(np.float32))


@pytest.mark.parametrize("scale", [2.0 ** -10, 1.0, 37.0, 3.0e4, 1.0e7])
def test_silhouette_grazing_rays_at_the_filter_margin(ra, oracle, golden_scenes, scale):
    """Aimed at the filters' margins: views along sphere silhouettes with neighbouring rays ulps apart, at scene scales
    2^-10 .. 10^7, eyes near and far, both filters, full-scan validation on.  A filter that dropped a grazing hit, or an
    exact test that resolved one differently from the reference, shows as a filter miss or as a primary-hit map that
    differs from the oracle's."""
    s = np.float32(scale)
    table = golden_scenes["book1"].copy()
    table[:, 0:4] *= s
    cols, rows, spp = 192, 128, 1
    # (sphere index, eye): the big glass and metal balls from the book camera, a small sphere seen from far off-axis, the ground
    # sphere (r = 1000) from just above it, and a small sphere from an eye almost on its surface
    views = [(483, (13.0, 2.0, 3.0)), (484, (13.0, 2.0, 3.0)), (100, (300.0, 150.0, -220.0)), (0, (3.0, 0.7, 9.0)),
             (200, tuple(table[200, 0:3] / s + np.array([0.21, 0.05, 0.02])))]
    near = 0
    for kind in (ra.FILTER_MFMA, ra.FILTER_VALU, ra.FILTER_GRID):
        ctx = ra.Context(0)
        try:
            ctx.set_filter(kind)
            ctx.upload_scene(table)
            for idx, eye in views:
                args = _grazing_view(table[idx, 0:3].astype(np.float64), float(table[idx, 3]), np.array(eye, dtype=np.float64) * float(s), cols, rows)
                ctx.set_camera(ra.Camera(*args))
                primary, st = ctx.render(cols, rows, 0, spp, 3, 5, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
                accum = ctx.read_accum()
                assert st["filter_misses"] == 0, (scale, kind, idx, st)
                o_accum, o_primary, _ = oracle.render(table, oracle.camera(*args), rows, cols, 0, spp, 3, 5, want_primary=True)
                assert (primary == o_primary).all(), (scale, kind, idx, int((primary != o_primary).sum()))
                same = (bits(accum) == bits(o_accum)) | (np.isnan(accum) & np.isnan(o_accum))
                assert same.all()
                # the silhouette of the aimed-at sphere is inside the frame: both sides of it are populated
                on = float((primary == idx).mean())
                if 0.02 < on < 0.98:
                    near += 1
        finally:
            ctx.close()
    assert near >= 9, near          # most views really straddle their silhouette (the rest are occluded or clipped by rounding)


def test_off_axis_sphere_at_the_f16_range_limit(ra, oracle):
    """The corner the round-1 range guard left open: a sphere at (M, M, M) with M just inside the
    per-component bound, where a bounce origin on the far side has |o|^2 near 3 M^2 and the per-ray constant of the MFMA
    filter would overflow f16.  The guard now bounds the norm; glass keeps rays bouncing inside, the full scan checks."""
    M = np.float32(22.4)
    rows_ = [[M, M, M, 0.3 * M, ra.MATERIAL_GLASS, 1, 1, 1, 0.0, 1.5],
             [M, M - 0.31 * M - 1000.0, M, 1000.0, ra.MATERIAL_DIFFUSE, 0.5, 0.5, 0.5, 1.0, 1.0],
             [M + 0.7 * M, M, M - 0.2 * M, 0.25 * M, ra.MATERIAL_METAL, 0.8, 0.7, 0.6, 0.0, 1.0],
             [M - 0.5 * M, M + 0.1 * M, M + 0.6 * M, 0.2 * M, ra.MATERIAL_GLASS, 1, 1, 1, 0.0, 1.5]]
    table = np.array(rows_, dtype=np.float32)
    cols, rows, spp = 160, 120, 4
    eye = np.array([M + 1.2 * M, M + 0.3 * M, M + 1.2 * M], dtype=np.float32)
    args = (40.0, np.float32(cols) / np.float32(rows), 0.0, np.float32(M), eye, (0, 1, 0), (M, M, M))
    for kind in (ra.FILTER_MFMA, ra.FILTER_VALU):
        ctx = ra.Context(0)
        try:
            ctx.set_filter(kind)
            ctx.upload_scene(table)
            ctx.set_camera(ra.Camera(*args))
            primary, st = ctx.render(cols, rows, 0, spp, 50, 3, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
            accum = ctx.read_accum()
        finally:
            ctx.close()
        assert st["filter_misses"] == 0 and st["rays"] > 2 * st["samples"]      # paths really bounce (the glass ball fills a good part of the frame)
        o_accum, o_primary, _ = oracle.render(table, oracle.camera(*args), rows, cols, 0, spp, 50, 3, want_primary=True)
        assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all()
        assert float((primary == 0).mean()) > 0.05


@pytest.mark.parametrize("factor", [2.0 ** -10, 1.0, 3.0e4, 1.0e7])
def test_grid_filter_scale_sweep_and_big_spheres(ra, oracle, golden_scenes, factor):
    """RTX_FILTER_GRID: the uniform grid at very different scene sizes, with the ground sphere on the
    big list, against the oracle bit for bit and with the full-scan validation on; its cell slack is relative to the scene."""
    table = golden_scenes["scaled"].copy()
    f = np.float32(factor)
    table[:, 0:4] *= f
    cols, rows, spp = 96, 64, 2
    pos, look = np.array([13, 2, 3], dtype=np.float32) * f, np.zeros(3, dtype=np.float32)
    args = (20.0, np.float32(cols) / np.float32(rows), np.float32(0.1) * f, np.float32(10.0) * f, pos, (0, 1, 0), look)
    ctx = ra.Context(0)
    try:
        ctx.set_filter(ra.FILTER_GRID)
        ctx.upload_scene(table)
        ctx.set_camera(ra.Camera(*args))
        primary, st = ctx.render(cols, rows, 0, spp, 50, 5, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
        accum = ctx.read_accum()
    finally:
        ctx.close()
    assert st["filter_misses"] == 0
    assert st["candidates"] < 0.05 * st["sphere_tests"]            # a few dozen exact tests per ray instead of 1601
    o_accum, o_primary, o_st = oracle.render(table, oracle.camera(*args), rows, cols, 0, spp, 50, 5, want_primary=True)
    same = (bits(accum) == bits(o_accum)) | (np.isnan(accum) & np.isnan(o_accum))
    assert (primary == o_primary).all() and same.all()
    assert st["rays"] == o_st["rays"]


def test_grid_filter_degenerate_tables(ra, oracle):
    """No sphere, one sphere, identical spheres, all spheres huge (nothing fits a grid), a sphere of radius 0 and a NaN centre:
    the grid filter must agree with the oracle wherever the other filters do."""
    cols, rows, spp = 48, 32, 2
    cases = {
        "empty": np.zeros((0, 10), dtype=np.float32),
        "one": np.array([[0, 0, 0, 1.5, 1, .5, .5, .5, 1, 1]], dtype=np.float32),
        "twins": np.array([[0, 0, 0, 1.5, 3, 1, 1, 1, 0, 1.5]] * 5, dtype=np.float32),
        "huge": np.array([[0, -1000, 0, 1000, 1, .5, .5, .5, 1, 1], [0, 1000.5, 0, 999, 2, .8, .8, .8, 0, 1], [2000, 0, 0, 1990, 3, 1, 1, 1, 0, 1.5]], dtype=np.float32),
        "zero_radius": np.array([[0, 0, 0, 0.0, 1, .5, .5, .5, 1, 1], [1, 0, 0, 0.5, 2, .5, .5, .5, 0.2, 1], [0, -100.5, 0, 100, 1, .3, .6, .3, 1, 1]], dtype=np.float32),
    }
    rng = np.random.default_rng(8)
    line = np.zeros((300, 10), dtype=np.float32)                     # a one-dimensional grid
    line[:, 0] = np.linspace(-30, 30, 300); line[:, 3] = 0.08; line[:, 4] = 1; line[:, 5:8] = 0.6; line[:, 9] = 1
    cases["line"] = line
    cloud = np.zeros((400, 10), dtype=np.float32)                    # a three-dimensional one, radii over two decades, some negative
    cloud[:, 0:3] = rng.uniform(-4, 4, size=(400, 3)); cloud[:, 3] = 10.0 ** rng.uniform(-2.5, -0.5, 400) * rng.choice([1, 1, 1, -1], 400)
    cloud[:, 4] = rng.integers(1, 4, 400); cloud[:, 5:8] = 0.7; cloud[:, 8] = 0.1; cloud[:, 9] = 1.5
    cases["cloud"] = cloud
    shifted = cloud.copy()                                           # the same far from the coordinate origin: the slack follows the coordinates
    shifted[:, 0:3] += np.float32(3.0e4)
    cases["shifted"] = shifted
    args = (40.0, np.float32(cols) / np.float32(rows), 0.05, 6.0, np.array([3, 1.5, 5], dtype=np.float32), (0, 1, 0), (0, 0, 0))
    for name, table in cases.items():
        if name == "shifted":
            args = (40.0, np.float32(cols) / np.float32(rows), 0.05, 6.0, np.array([3, 1.5, 5], dtype=np.float32) + np.float32(3.0e4), (0, 1, 0), (3.0e4, 3.0e4, 3.0e4))
        ctx = ra.Context(0)
        try:
            ctx.set_filter(ra.FILTER_GRID)
            ctx.upload_scene(table)
            ctx.set_camera(ra.Camera(*args))
            primary, st = ctx.render(cols, rows, 0, spp, 50, 9, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
            accum = ctx.read_accum()
        finally:
            ctx.close()
        assert st["filter_misses"] == 0, name
        o_accum, o_primary, _ = oracle.render(table, oracle.camera(*args), rows, cols, 0, spp, 50, 9, want_primary=True)
        same = (bits(accum) == bits(o_accum)) | (np.isnan(accum) & np.isnan(o_accum))
        assert (primary == o_primary).all() and same.all(), name


def test_auto_filter_picks_by_table_size(ra, golden_scenes):
    """RTX_FILTER_AUTO is the MFMA filter up to 256 spheres and the grid from there on: same image either way, and the
    number of exact tests tells which one ran; the grid's density knob changes which spheres reach the exact test and nothing else."""
    cols, rows, spp = 96, 64, 2
    args = (20.0, np.float32(cols) / np.float32(rows), 0.1, 10.0, (13, 2, 3), (0, 1, 0), (0, 0, 0))

    def run(table, kind, density=None):
        ctx = ra.Context(0)
        try:
            ctx.set_filter(kind)
            if density is not None:
                ctx.set_tuning("grid_density_pct", density)
            ctx.upload_scene(table)
            ctx.set_camera(ra.Camera(*args))
            _, st = ctx.render(cols, rows, 0, spp, 50, 3, chunk_spp=spp)
            return ctx.read_accum(), st
        finally:
            ctx.close()

    for table, expect in ((ra.random_scene(1, 5), ra.FILTER_MFMA), (golden_scenes["book1"], ra.FILTER_GRID), (golden_scenes["scaled"], ra.FILTER_GRID)):
        assert (len(table) > 256) == (expect == ra.FILTER_GRID)
        a_auto, s_auto = run(table, ra.FILTER_AUTO)
        a_ref, s_ref = run(table, expect)
        other = ra.FILTER_GRID if expect == ra.FILTER_MFMA else ra.FILTER_MFMA
        a_other, s_other = run(table, other)
        assert (bits(a_auto) == bits(a_ref)).all() and (bits(a_auto) == bits(a_other)).all()
        assert s_auto["candidates"] == s_ref["candidates"] != s_other["candidates"]       # which one ran: the number of exact tests
    table = golden_scenes["scaled"]
    a_coarse, s_coarse = run(table, ra.FILTER_GRID, density=400)
    a_fine, s_fine = run(table, ra.FILTER_GRID, density=12)
    assert (bits(a_coarse) == bits(a_fine)).all() and s_coarse["candidates"] != s_fine["candidates"]
    ctx = ra.Context(0)
    try:
        with pytest.raises(Exception):
            ctx.set_tuning("grid_density_pct", 0)
        with pytest.raises(Exception):
            ctx.set_filter(4)
    finally:
        ctx.close()


@pytest.mark.parametrize("case", ["across_the_field", "tiny_spheres", "from_far_away", "from_very_far_away"])
def test_grid_filter_reproduces_the_far_field_noise(ra, case):
    """Far from the ray's origin the reference's fp32 discriminant is noise of size 2^-19 (L^2 + r^2): a ray can "hit" a sphere
    that it passes at a distance.  The grid registers spheres with that blur, tier by tier (rtx_host.cpp, build_grid), and rays
    that outrun the outermost tier scan the whole table: with the full-scan validation on, no ray may disagree -- looking
    across a wide field at grazing height, with spheres of radius 1e-3 and 0, and from cameras 3 and 300 field widths away."""
    table = ra.random_scene(3, 40).copy()                      # 6 400 small spheres on a 80 x 80 field + the ground and the three large ones
    cols, rows, spp = 160, 100, 2
    pos, look, fov = (-38.0, 0.6, 0.5), (40.0, 0.2, 0.0), 30.0
    if case == "tiny_spheres":
        table[4::3, 3] = 1.0e-3
        table[5::7, 3] = 0.0
    elif case == "from_far_away":
        pos, fov = (-250.0, 30.0, 20.0), 12.0
    elif case == "from_very_far_away":
        pos, fov = (-25000.0, 3000.0, 2000.0), 0.12
    ctx = ra.Context(0)
    try:
        ctx.upload_scene(table)
        ctx.set_camera(ra.Camera(fov, np.float32(cols) / np.float32(rows), 0.0, 10.0, pos, (0, 1, 0), look))
        out = {}
        for kind in (ra.FILTER_GRID, ra.FILTER_VALU):
            ctx.set_filter(kind)
            primary, st = ctx.render(cols, rows, 0, spp, 6, 11, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
            assert st["filter_misses"] == 0, (case, kind, st)
            out[kind] = (primary, ctx.read_accum(), st)
        # the production kernel of the grid (no validation: another instantiation of the template) gives the same frame
        ctx.set_filter(ra.FILTER_GRID)
        primary, st = ctx.render(cols, rows, 0, spp, 6, 11, chunk_spp=spp, want_primary=True)
        plain = ctx.read_accum()
        same = (bits(plain) == bits(out[ra.FILTER_GRID][1])) | (np.isnan(plain) & np.isnan(out[ra.FILTER_GRID][1]))
        assert (primary == out[ra.FILTER_GRID][0]).all() and same.all() and st["rays"] == out[ra.FILTER_GRID][2]["rays"]
    finally:
        ctx.close()
    (p_g, a_g, s_g), (p_v, a_v, s_v) = out[ra.FILTER_GRID], out[ra.FILTER_VALU]
    same = (bits(a_g) == bits(a_v)) | (np.isnan(a_g) & np.isnan(a_v))
    assert (p_g == p_v).all() and same.all() and s_g["rays"] == s_v["rays"]
    assert float((p_g > 3).mean()) > 0.02                      # small spheres are in view
    if case in ("across_the_field", "tiny_spheres"):
        assert s_g["candidates"] < 0.05 * s_g["sphere_tests"]  # served by the grid ...
    if case == "from_very_far_away":
        assert s_g["candidates"] > 0.3 * s_g["sphere_tests"]   # ... and here by the fallback scan (primary rays; the bounces use the grid)


@pytest.mark.parametrize("scene,cols,rows,spp", [("book1", 301, 199, 6), ("glass", 200, 120, 6), ("scaled", 160, 100, 4)])
def test_parked_grid_walks_change_nothing(ra, oracle, golden_scenes, scene, cols, rows, spp):
    """Lane decoupling (tuning knob walk_park): the walks of a wave's last rays are parked in LDS while the wave shades the others and
    go on in the next iteration.  Whatever the threshold -- never, a single ray, the default, half the wave -- every ray still meets
    the full scan's hit (RTX_FLAG_VALIDATE_FILTER) and the frame equals the oracle's bit for bit; so do the counters."""
    table = golden_scenes[scene]
    o_accum, o_primary, o_st = oracle.render(table, oracle.book_camera(cols, rows), rows, cols, 0, spp, 50, 21, want_primary=True)
    tests = None
    for park in (0, 1, None, 32):
        ctx = ra.Context(0)
        ctx.set_filter(ra.FILTER_GRID)
        if park is not None:
            ctx.set_tuning("walk_park", park)
        ctx.upload_scene(table)
        ctx.set_camera(ra.book_camera(cols, rows))
        primary, st = ctx.render(cols, rows, 0, spp, 50, 21, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
        assert st["filter_misses"] == 0, (park, st)
        accum = ctx.read_accum()
        primary2, st2 = ctx.render(cols, rows, 0, spp, 50, 21, chunk_spp=spp, want_primary=True)          # the production instantiation
        assert (primary2 == primary).all() and (bits(ctx.read_accum()) == bits(accum)).all()
        ctx.close()
        assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all(), park
        for k in ("samples", "rays", "term_miss", "term_absorb", "term_depth"):
            assert st[k] == o_st[k] == st2[k], (park, k)
        tests = st["candidates"] if tests is None else tests
        assert st["candidates"] == tests == st2["candidates"]         # the same exact tests, only in another iteration


def test_walk_park_knob_is_checked(ra):
    ctx = ra.Context(0)
    with pytest.raises(Exception):
        ctx.set_tuning("walk_park", 33)
    with pytest.raises(Exception):
        ctx.set_tuning("walk_park", -1)
    ctx.close()


def test_grid_cell_with_more_entries_than_its_record_counts(ra, oracle):
    """A grid cell's record counts the entries of its two inner tiers in 16 bits.  A table that puts more spheres than that through
    one cell (here 66 000 specks in a cluster a fifth of a cell wide, in a field of 1 500 spread spheres) gets no grid: every
    sphere goes on the list that each ray tests in full -- the reference's own loop -- and the image still equals the oracle's."""
    rng = np.random.default_rng(5)
    n_cluster, n_field = 66000, 1500
    t = np.zeros((n_cluster + n_field + 1, 10), dtype=np.float32)
    t[0, 0:4] = (0, -1000, 0, 1000)
    t[1:1 + n_field, 0] = rng.uniform(-20, 20, n_field); t[1:1 + n_field, 1] = 0.2; t[1:1 + n_field, 2] = rng.uniform(-20, 20, n_field)
    t[1:1 + n_field, 3] = 0.2
    t[1 + n_field:, 0:3] = np.array([1.0, 0.3, 0.5], dtype=np.float32) + rng.uniform(-0.05, 0.05, size=(n_cluster, 3)).astype(np.float32)
    t[1 + n_field:, 3] = 0.01
    t[:, 4] = 1; t[:, 5:8] = 0.6; t[:, 8] = 0; t[:, 9] = 1
    cols, rows, spp = 40, 24, 1
    args = (5.0, np.float32(cols) / np.float32(rows), 0.0, 10.0, np.array([6, 2, 3], dtype=np.float32), (0, 1, 0), (1, 0.3, 0.5))
    ctx = ra.Context(0)
    try:
        ctx.set_filter(ra.FILTER_GRID)
        ctx.upload_scene(t)
        ctx.set_camera(ra.Camera(*args))
        primary, st = ctx.render(cols, rows, 0, spp, 8, 4, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
        accum = ctx.read_accum()
    finally:
        ctx.close()
    assert st["filter_misses"] == 0
    assert st["candidates"] == st["rays"] * len(t)                   # no grid: every ray met every sphere
    o_accum, o_primary, o_st = oracle.render(t, oracle.camera(*args), rows, cols, 0, spp, 8, 4, want_primary=True)
    assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all() and st["rays"] == o_st["rays"]
    assert (primary >= 1 + n_field).any()                           # the cluster is in the picture
