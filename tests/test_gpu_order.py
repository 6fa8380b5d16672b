"""Cost-ordered work lists (rtx_last_launch_ordered): a launch of a frame the context has traced before hands its work out
most expensive strip first.  The order must never change a result: every ordered launch is compared with the oracle, which
sums a pixel's samples in the chunks the library reports (rtx_last_chunk_plan), bit for bit.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import raytracing_amd
    raytracing_amd.lib()
    return raytracing_amd


@pytest.fixture()
def ctx(ra):
    c = ra.Context(0)
    yield c
    c.close()


def _oracle_check(ctx, oracle, table, cols, rows, begin, spp, depth, seed, primary, st):
    chunk_spp, starts = ctx.last_chunk_plan()
    assert chunk_spp == 0 and starts[0] == 0 and starts[-1] == spp
    o_accum, o_primary, o_st = oracle.render(table, oracle.book_camera(cols, rows), rows, cols, begin, spp, depth, seed, want_primary=True, chunk_starts=starts)
    assert (primary == o_primary).all() and (bits(ctx.read_accum()) == bits(o_accum)).all()
    for k in ("samples", "rays", "term_miss", "term_absorb", "term_depth"):
        assert st[k] == o_st[k], k
    return starts


@pytest.mark.parametrize("filt", ["mfma", "grid", "valu"])
@pytest.mark.parametrize("cols,rows", [(301, 199), (128, 64)])       # 59 899 pixels: the last strip reaches past the image; 8192: whole strips
def test_second_launch_of_a_frame_is_ordered_and_bit_exact(ra, ctx, oracle, golden_scenes, filt, cols, rows):
    table = golden_scenes["book1"]
    spp, depth, seed = 24, 8, 7
    ctx.set_filter({"mfma": ra.FILTER_MFMA, "grid": ra.FILTER_GRID, "valu": ra.FILTER_VALU}[filt])
    ctx.set_tuning("order_share_pct", 400)         # small frames: let the rule pick chunks of >= 3 samples at this sample count
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    primary, st = ctx.render(cols, rows, 0, spp, depth, seed, want_primary=True)
    assert not ctx.last_launch_ordered()                                 # nothing is known about the frame yet
    first = ctx.read_accum()
    _oracle_check(ctx, oracle, table, cols, rows, 0, spp, depth, seed, primary, st)
    primary2, st2 = ctx.render(cols, rows, 0, spp, depth, seed, want_primary=True)
    if cols * rows * spp < 400000:
        assert not ctx.last_launch_ordered()                             # too little work per lane for chunks of three samples
        return
    assert ctx.last_launch_ordered()
    starts = _oracle_check(ctx, oracle, table, cols, rows, 0, spp, depth, seed, primary2, st2)
    assert 2 <= len(starts) - 1 <= 16 and (primary2 == primary).all() and st2["rays"] == st["rays"]
    np.testing.assert_allclose(ctx.read_accum(), first, rtol=1e-5, atol=1e-5)          # another chunking: float accuracy
    # another sample range of the same frame (a progressive pass): ordered by the costs of the previous launch
    primary3, st3 = ctx.render(cols, rows, spp, spp + 5, depth, seed, want_primary=True)
    assert ctx.last_launch_ordered()
    _oracle_check(ctx, oracle, table, cols, rows, spp, spp + 5, depth, seed, primary3, st3)
    # caller-chosen chunks are never ordered; ordering can be switched off
    ctx.render(cols, rows, 0, spp, depth, seed, chunk_spp=6)
    assert not ctx.last_launch_ordered()
    ctx.set_tuning("order", 0)
    ctx.render(cols, rows, 0, spp, depth, seed)
    assert not ctx.last_launch_ordered()


def test_a_new_camera_scene_or_size_starts_unordered(ra, ctx, oracle, golden_scenes):
    table, cols, rows, spp, depth, seed = golden_scenes["book1"], 301, 199, 24, 8, 3
    ctx.set_tuning("order_share_pct", 400)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    ctx.render(cols, rows, 0, spp, depth, seed)
    ctx.render(cols, rows, 0, spp, depth, seed)
    assert ctx.last_launch_ordered()
    cam2 = ra.Camera(25.0, cols / rows, 0.05, 9.0, (10, 3, 4), (0, 1, 0), (0, 0.5, 0))
    ctx.set_camera(cam2)
    ctx.render(cols, rows, 0, spp, depth, seed)
    assert not ctx.last_launch_ordered()                                 # costs of the other view say nothing about this one
    ctx.render(cols, rows, 0, spp, depth, seed)
    assert ctx.last_launch_ordered()
    ctx.upload_scene(golden_scenes["glass"])
    ctx.render(cols, rows, 0, spp, depth, seed)
    assert not ctx.last_launch_ordered()
    ctx.render(cols, rows, 0, spp, depth, seed)
    assert ctx.last_launch_ordered()
    ctx.render(rows, cols, 0, spp, depth, seed)                          # transposed size
    assert not ctx.last_launch_ordered()
    # the ordered launch of the glass scene against the oracle
    primary, st = ctx.render(rows, cols, 0, spp, depth, seed, want_primary=True)
    assert ctx.last_launch_ordered()
    chunk_spp, starts = ctx.last_chunk_plan()
    o_accum, o_primary, o_st = oracle.render(golden_scenes["glass"], oracle.camera_from_array(cam2.as_array()), cols, rows, 0, spp, depth, seed, want_primary=True, chunk_starts=starts)
    assert (primary == o_primary).all() and (bits(ctx.read_accum()) == bits(o_accum)).all() and st["rays"] == o_st["rays"]


def test_share_of_one_of_eight_gpus_on_the_bench_frame(ra, ctx, golden_scenes):
    """BASELINE's 500 spp split over 8 GPUs is 62-63 samples per rank on the 1200x800 frame: the second launch is ordered with at
    most 16 chunks (the first, with the geometric plan, has more) and, where both use the same chunks, equals an unordered
    launch bit for bit; counters are identical."""
    cols, rows, depth, seed = 1200, 800, 50, 1
    ctx.set_filter(ra.FILTER_MFMA)                                       # bench.py's path (the chunk sizes depend on the kernel's resident lanes)
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(cols, rows))
    spp = 66
    _, st = ctx.render(cols, rows, 0, spp, depth, seed)
    n_first = len(ctx.last_chunk_plan()[1]) - 1
    assert not ctx.last_launch_ordered()
    _, st2 = ctx.render(cols, rows, 0, spp, depth, seed)
    assert ctx.last_launch_ordered()
    starts = ctx.last_chunk_plan()[1]
    sizes = np.diff(starts.astype(np.int64))
    assert len(sizes) <= 16 and len(sizes) < n_first and sizes.min() >= 3
    ordered = ctx.read_accum()
    for k in ("samples", "rays", "sphere_tests", "term_miss", "term_absorb", "term_depth", "candidates"):
        assert st2[k] == st[k], k
    assert len(set(sizes.tolist())) == 1                                 # 66 samples: 11 chunks of 6
    ctx.render(cols, rows, 0, spp, depth, seed, chunk_spp=int(sizes[0]), want_stats=False)
    assert not ctx.last_launch_ordered()
    assert (bits(ctx.read_accum()) == bits(ordered)).all()
