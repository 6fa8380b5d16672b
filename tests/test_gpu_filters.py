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
    for kind in (ra.FILTER_VALU, ra.FILTER_MFMA):
        ctx = ra.Context(0)
        ctx.set_filter(kind)
        ctx.upload_scene(golden_scenes[scene])
        ctx.set_camera(ra.book_camera(cols, rows))
        primary, st = ctx.render(cols, rows, 0, spp, 50, 77, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=spp, want_primary=True)
        assert st["filter_misses"] == 0, (kind, st)
        assert st["candidates"] < 0.02 * st["sphere_tests"]       # the filter passes only a small share on to the exact test
        results[kind] = (primary, ctx.read_accum(), st)
        ctx.close()
    (p0, a0, s0), (p1, a1, s1) = results[ra.FILTER_VALU], results[ra.FILTER_MFMA]
    assert (p0 == p1).all() and (bits(a0) == bits(a1)).all()
    for k in ("samples", "rays", "term_miss", "term_absorb", "term_depth"):
        assert s0[k] == s1[k]


def test_far_origins_and_large_offsets(ra, oracle, golden_scenes):
    """A scene pushed far from the coordinate origin stresses the MFMA filter's absolute margins (|C|^2 + |o|^2)."""
    table = golden_scenes["book1"].copy()
    shift = np.array([300.0, 0.0, -200.0], dtype=np.float32)
    table[:, 0:3] += shift
    cols, rows, spp = 160, 100, 2
    pos = np.array([13, 2, 3], dtype=np.float32) + shift
    cam = ra.Camera(20.0, np.float32(cols) / np.float32(rows), 0.1, 10.0, pos, (0, 1, 0), shift)
    ocam = oracle.camera(20.0, np.float32(cols) / np.float32(rows), 0.1, 10.0, pos, (0, 1, 0), shift)
    for kind in (ra.FILTER_VALU, ra.FILTER_MFMA):
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
