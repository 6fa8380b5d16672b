"""HIP backend (through the C ABI) against the oracle and the reference-made golden frames.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import raytracing_amd
    raytracing_amd.lib()
    return raytracing_amd


@pytest.fixture(scope="module")
def ctx(ra):
    c = ra.Context(0)
    yield c
    c.close()


def _hip_frame(ra, ctx, table, f, chunk):
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(f["cols"], f["rows"]))
    flags = ra.FLAG_RECURSIVE if f["recursive"] else 0
    primary, st = ctx.render(f["cols"], f["rows"], f["spp_begin"], f["spp_count"], f["max_depth"], f["seed"], flags=flags, chunk_spp=chunk, want_primary=True)
    return ctx.read_accum(), primary, st


@pytest.mark.parametrize("filt", ["mfma", "valu", "grid"])
def test_golden_frames_bit_exact(ra, ctx, golden_scenes, golden_frames, filt):
    """Primary-hit indices, colour sums and 8-bit framebuffers equal what the compiled reference produced (the -r frame
    included: raytracer.cpp:213-240 multiplies the attenuations innermost first), whichever way the candidates are found."""
    ctx.set_filter({"mfma": ra.FILTER_MFMA, "valu": ra.FILTER_VALU, "grid": ra.FILTER_GRID}[filt])
    for name, f in golden_frames.items():
        accum, primary, st = _hip_frame(ra, ctx, golden_scenes[f["scene"]], f, f["spp_count"])
        if "primary" in f:
            assert (primary == f["primary"]).all(), name
        if "accum" in f:
            assert (bits(accum) == bits(f["accum"])).all(), name
        if "fb" in f:
            assert (ctx.resolve(f["spp_count"]) == f["fb"]).all(), name
        assert st["samples"] == f["rows"] * f["cols"] * f["spp_count"]
        assert st["sphere_tests"] == st["rays"] * f["n"]
        assert st["term_miss"] + st["term_absorb"] + st["term_depth"] == st["samples"]
        if not f["recursive"]:
            assert st["rays"] == f["scatter_events"] + st["term_miss"]


@pytest.mark.parametrize("cols,rows,spp,depth,chunk", [(97, 61, 3, 50, 0), (64, 64, 8, 50, 3), (33, 7, 1, 1, 0), (128, 96, 4, 5, 2)])
def test_against_oracle(ra, ctx, oracle, golden_scenes, cols, rows, spp, depth, chunk):
    table = golden_scenes["book1"]
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    eff_chunk = chunk if chunk else spp
    primary, st = ctx.render(cols, rows, 0, spp, depth, 42, chunk_spp=eff_chunk, want_primary=True)
    accum = ctx.read_accum()
    o_accum, o_primary, o_st = oracle.render(table, oracle.book_camera(cols, rows), rows, cols, 0, spp, depth, 42, chunk_spp=eff_chunk, want_primary=True)
    assert (primary == o_primary).all()
    assert (bits(accum) == bits(o_accum)).all()
    for k in ("samples", "rays", "sphere_tests", "term_miss", "term_absorb", "term_depth"):
        assert st[k] == o_st[k], k
    assert (ctx.resolve(spp) == oracle.resolve(o_accum, rows, cols, spp)).all()


@pytest.mark.parametrize("cols,rows,spp,depth", [(97, 61, 3, 50), (64, 48, 4, 2), (40, 30, 2, 0)])
def test_recursive_against_oracle(ra, ctx, oracle, golden_scenes, cols, rows, spp, depth):
    """-r (raytracer.cpp:213-240): absorbed and over-depth paths are black, the ray at depth == max_depth is still traced,
    and the product of attenuations is formed innermost first."""
    table = golden_scenes["book1"]
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    primary, st = ctx.render(cols, rows, 0, spp, depth, 9, flags=ra.FLAG_RECURSIVE, chunk_spp=spp, want_primary=True)
    accum = ctx.read_accum()
    o_accum, o_primary, o_st = oracle.render(table, oracle.book_camera(cols, rows), rows, cols, 0, spp, depth, 9, chunk_spp=spp, recursive=True, want_primary=True)
    assert (primary == o_primary).all()
    assert (bits(accum) == bits(o_accum)).all()
    for k in ("samples", "rays", "sphere_tests", "term_miss", "term_absorb", "term_depth"):
        assert st[k] == o_st[k], k


@pytest.mark.parametrize("name", ["book1_120x80_spp1024", "glass_120x80_spp1024", "book1_120x80_spp1024_recursive", "scaled_120x80_spp1024"])
def test_psnr_against_unmodified_reference_render(ra, ctx, oracle, golden_scenes, name):
    """BASELINE's image bar: the unmodified renderScene of the reference (its own mt19937 stream, 1024 spp, fixtures made by
    oracle/gen_fixtures.py / gen_refrenders.py; iterative and -r) and the HIP backend (Philox, 8192 spp) are independent
    Monte Carlo estimates of one image and must agree to >= 40 dB."""
    import json, os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "refrender.json")) as fjs:
        m = json.load(fjs)[name]
    ref = np.load(os.path.join(GOLDEN, "refrender_" + name + ".npz"))["fb"]
    spp = 8192
    ctx.upload_scene(golden_scenes[m["scene"]])
    ctx.set_camera(ra.book_camera(m["cols"], m["rows"]))
    ctx.render(m["cols"], m["rows"], 0, spp, m["max_depth"], 11, flags=ra.FLAG_RECURSIVE if m.get("recursive") else 0, want_stats=False)
    psnr, rmse = oracle.psnr_rgb8(ctx.resolve(spp), ref)
    assert psnr >= 40.0, (name, psnr, rmse)


@pytest.mark.parametrize("scene,filt", [("book1", "mfma"), ("glass", "mfma"), ("book1", "grid"), ("scaled", "grid"), ("scaled", "mfma")])
def test_benchmark_frame_primary_hits_and_image(ra, ctx, oracle, golden_scenes, scene, filt):
    """north_star's first bar on the BENCHMARK frame itself: 1200x800, depth 50 -- the sphere index of
    every primary hit (raytracer.cpp:285-304, lowest index on ties), the colour sums, the counters and the 8-bit
    framebuffer equal the oracle's, which runs the same 960 000 paths on the host cores."""
    cols, rows, spp = 1200, 800, 1
    table = golden_scenes[scene]
    ctx.set_filter(ra.FILTER_GRID if filt == "grid" else ra.FILTER_MFMA)      # the production (non-validating) kernels of both
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    primary, st = ctx.render(cols, rows, 0, spp, 50, 1, chunk_spp=spp, want_primary=True)
    accum = ctx.read_accum()
    o_accum, o_primary, o_st = oracle.render(table, oracle.book_camera(cols, rows), rows, cols, 0, spp, 50, 1, want_primary=True, nthreads=16)
    assert (primary == o_primary).all()
    assert (bits(accum) == bits(o_accum)).all()
    for k in ("samples", "rays", "sphere_tests", "term_miss", "term_absorb", "term_depth"):
        assert st[k] == o_st[k], k
    assert (ctx.resolve(spp) == oracle.resolve(o_accum, rows, cols, spp)).all()
    assert 0.3 < float((primary != 0xFFFFFFFF).mean()) < 1.0 and len(np.unique(primary)) > 50
