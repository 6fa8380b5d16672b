"""Edge cases and size-independent properties of the HIP path (through the C ABI).  Needs an MI355X."""
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


def _both(ra, ctx, oracle, table, cols, rows, spp_begin, spp, depth, seed, chunk=0):
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    eff = chunk if chunk else spp
    primary, st = ctx.render(cols, rows, spp_begin, spp, depth, seed, chunk_spp=eff, want_primary=True)
    accum = ctx.read_accum()
    o_accum, o_primary, o_st = oracle.render(table, oracle.book_camera(cols, rows), rows, cols, spp_begin, spp, depth, seed, chunk_spp=eff, want_primary=True)
    return accum, primary, st, o_accum, o_primary, o_st


def test_empty_scene(ra, ctx, oracle):
    """n = 0: every ray ends on the background (raytracer.cpp:266-269)."""
    table = np.zeros((0, 10), dtype=np.float32)
    accum, primary, st, o_accum, o_primary, o_st = _both(ra, ctx, oracle, table, 40, 30, 0, 3, 50, 1)
    assert (primary == 0xFFFFFFFF).all() and (bits(accum) == bits(o_accum)).all()
    assert st["rays"] == st["samples"] == st["term_miss"] == 40 * 30 * 3 and st["sphere_tests"] == 0


@pytest.mark.parametrize("n", [1, 31, 32, 33, 65])
def test_ragged_table_sizes(ra, ctx, oracle, golden_scenes, n):
    """Sphere counts around the 32-sphere filter block (padding must never hit)."""
    book = golden_scenes["book1"]
    table = np.concatenate([book[:max(0, n - 3)], book[-min(3, n):]])[:n] if n > 3 else book[-n:]
    assert len(table) == n
    accum, primary, st, o_accum, o_primary, o_st = _both(ra, ctx, oracle, table, 64, 40, 0, 2, 50, 3)
    assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all()
    assert primary[primary != 0xFFFFFFFF].max(initial=0) < n


def test_one_pixel_and_thin_images(ra, ctx, oracle, golden_scenes):
    for cols, rows in [(1, 1), (1, 17), (257, 1), (3, 2)]:
        accum, primary, st, o_accum, o_primary, _ = _both(ra, ctx, oracle, golden_scenes["book1"], cols, rows, 0, 5, 50, 4)
        assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all()


def test_large_table_streams_from_l2(ra, ctx, oracle):
    """More spheres than the LDS-resident limit: the filter reads the table through L2 instead (BASELINE config 5 path)."""
    table = ra.random_scene(1, grid_half=36)          # ~5.1k spheres > 4096
    assert len(table) > 4096
    accum, primary, st, o_accum, o_primary, o_st = _both(ra, ctx, oracle, table, 40, 24, 0, 1, 50, 2)
    assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all()
    assert st["rays"] == o_st["rays"]


def test_glass_heavy_scene(ra, ctx, oracle, golden_scenes):
    """BASELINE config 4: mostly dielectric spheres, long paths, refraction branches."""
    accum, primary, st, o_accum, o_primary, o_st = _both(ra, ctx, oracle, golden_scenes["glass"], 80, 60, 0, 4, 50, 6, chunk=2)
    assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all()
    for k in ("rays", "term_miss", "term_absorb", "term_depth"):
        assert st[k] == o_st[k]
    assert st["rays"] / st["samples"] > 3.0


def test_depth_zero_and_one(ra, ctx, oracle, golden_scenes):
    table = golden_scenes["book1"]
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(20, 10))
    _, st = ctx.render(20, 10, 0, 3, 0, 1)
    assert (ctx.read_accum() == 3.0).all() and st["rays"] == 0          # the loop of raytracer.cpp:250 never runs
    assert (ctx.resolve(3) == 0xFFFFFF00).all()
    accum, primary, st, o_accum, o_primary, o_st = _both(ra, ctx, oracle, table, 20, 10, 0, 3, 1, 1)
    assert (bits(accum) == bits(o_accum)).all() and st["rays"] == st["samples"]


def test_sample_ranges_add_up(ra, ctx, golden_scenes):
    """Rendering [0,a) then accumulating [a,b) equals rendering [0,b) with chunk a (same summation order)."""
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(60, 40))
    ctx.render(60, 40, 0, 8, 50, 11, chunk_spp=4)
    whole = ctx.read_accum()
    ctx.render(60, 40, 0, 4, 50, 11, chunk_spp=4)
    ctx.render(60, 40, 4, 4, 50, 11, flags=ra.FLAG_ACCUMULATE, chunk_spp=4)
    assert (bits(ctx.read_accum()) == bits(whole)).all()


def test_deterministic_and_chunking_invariant_primary(ra, ctx, golden_scenes):
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(200, 120))
    p1, s1 = ctx.render(200, 120, 0, 6, 50, 5, chunk_spp=6, want_primary=True)
    a1 = ctx.read_accum()
    p2, s2 = ctx.render(200, 120, 0, 6, 50, 5, chunk_spp=6, want_primary=True)
    assert (bits(a1) == bits(ctx.read_accum())).all() and (p1 == p2).all()
    p3, s3 = ctx.render(200, 120, 0, 6, 50, 5, chunk_spp=1, want_primary=True)
    a3 = ctx.read_accum()
    assert (p1 == p3).all() and np.allclose(a1, a3, rtol=1e-5, atol=1e-6)
    for k in ("rays", "samples", "term_miss", "term_absorb", "term_depth"):
        assert s1[k] == s2[k] == s3[k]
    p4, _ = ctx.render(200, 120, 0, 6, 50, 6, chunk_spp=6, want_primary=True)
    assert (p1 != p4).any()                                            # another seed, another sample set


def test_debug_borders_and_resolve(ra, ctx, oracle, golden_scenes):
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(130, 70))
    ctx.render(130, 70, 0, 2, 50, 1, chunk_spp=2)
    accum = ctx.read_accum()
    assert (ctx.resolve(2, debug_tile=64) == oracle.resolve(accum, 70, 130, 2, debug_tile=64)).all()
    assert (ctx.resolve(2) == oracle.resolve(accum, 70, 130, 2)).all()


def test_write_accum_roundtrip_and_borrowed_buffer(ra, ctx, golden_scenes):
    import torch
    rng = np.random.default_rng(0)
    a = rng.random(50 * 20 * 3, dtype=np.float32)
    ctx.write_accum(50, 20, a)
    assert (bits(ctx.read_accum()) == bits(a)).all()
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(50, 20))
    ctx.render(50, 20, 0, 2, 50, 1, chunk_spp=2)
    own = ctx.read_accum()
    t = torch.zeros(50 * 20 * 3, dtype=torch.float32, device="cuda:0")
    ctx.use_accum_buffer(50, 20, t.data_ptr())
    ctx.render(50, 20, 0, 2, 50, 1, chunk_spp=2)
    ctx.sync()
    assert (bits(t.cpu().numpy()) == bits(own)).all()
    with pytest.raises(ra.RtxError):
        ctx.render(51, 20, 0, 2, 50, 1)                                 # the borrowed buffer has another size
    ctx.use_accum_buffer(0, 0, 0)


def test_checkpoint_resume_is_bit_identical(ra, ctx, golden_scenes, tmp_path):
    """Progressive rendering: 3 passes of 2 spp, with a save / load through a file and a NEW context
    in the middle, give the very floats of the uninterrupted 3 passes."""
    cols, rows, depth, seed = 80, 50, 50, 5
    table = golden_scenes["book1"]
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    for k in range(3):
        ctx.render(cols, rows, 2 * k, 2, depth, seed, flags=ra.FLAG_ACCUMULATE if k else 0, chunk_spp=2)
    straight = ctx.read_accum()
    ctx.render(cols, rows, 0, 2, depth, seed, chunk_spp=2)
    ck = tmp_path / "frame.accum"
    ctx.save_accum(ck, 2, seed, depth)
    other = ra.Context(0)
    try:
        other.upload_scene(table)
        other.set_camera(ra.book_camera(cols, rows))
        info = other.load_accum(ck)
        assert info == {"cols": cols, "rows": rows, "spp_done": 2, "seed": seed, "max_depth": depth, "flags": 0}
        for k in (1, 2):
            other.render(cols, rows, 2 * k, 2, depth, seed, flags=ra.FLAG_ACCUMULATE, chunk_spp=2)
        assert (other.read_accum().view(np.uint32) == straight.view(np.uint32)).all()
        with pytest.raises(ra.RtxError, match="checkpoint"):
            bad = tmp_path / "bad.accum"
            bad.write_bytes(b"not a checkpoint at all, just forty-four bytes..")
            other.load_accum(bad)
    finally:
        other.close()


def test_argument_errors(ra, ctx, golden_scenes):
    with pytest.raises(ra.RtxError, match="upload a scene"):
        ctx.render(8, 8, 0, 1, 5, 1)
    bad = golden_scenes["book1"].copy()
    bad[3, 4] = 7
    with pytest.raises(ra.RtxError, match="unknown material"):
        ctx.upload_scene(bad)
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(8, 8))
    with pytest.raises(ra.RtxError):
        ctx.render(8, 8, 0, 0, 5, 1)
    with pytest.raises(ra.RtxError):
        ctx.render(0, 8, 0, 1, 5, 1)
    with pytest.raises(ra.RtxError, match="ACCUMULATE"):
        ctx.render(9, 9, 0, 1, 5, 1, flags=ra.FLAG_ACCUMULATE)
    with pytest.raises(ra.RtxError):
        ra.Context(99)


def test_render_scene_hip_signature(ra, oracle, golden_scenes):
    """The drop-in wrapper with renderScene's argument list (raytracer.h:17) fills the caller's framebuffer."""
    table = golden_scenes["book1"]
    fb = np.full(48 * 32, 0xCCCCCCCC, dtype=np.uint32)                  # RayTracing.cpp:128 pre-fills with 0xCC
    rc = ra.render_scene_hip(ra.Scene.from_table(table), ra.book_camera(48, 32), 32, 48, fb, 2, 50, 1, 64, False, False, seed=5)
    assert rc == 0
    o_accum, _, _ = oracle.render(table, oracle.book_camera(48, 32), 32, 48, 0, 2, 50, 5)
    assert (fb == oracle.resolve(o_accum, 32, 48, 2)).all()


def test_full_size_properties(ra, ctx, golden_scenes):
    """BASELINE's frame size, 1200x800: counters, split additivity and the image statistics at 4 spp."""
    cols, rows, spp = 1200, 800, 4
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(cols, rows))
    _, st = ctx.render(cols, rows, 0, spp, 50, 1, chunk_spp=2)
    whole = ctx.read_accum()
    assert st["samples"] == cols * rows * spp
    assert st["term_miss"] + st["term_absorb"] + st["term_depth"] == st["samples"]
    assert st["sphere_tests"] == st["rays"] * 485
    assert 2.6 < st["rays"] / st["samples"] < 2.9                       # 2.71 rays per sample
    ctx.render(cols, rows, 0, 2, 50, 1, chunk_spp=2)
    ctx.render(cols, rows, 2, 2, 50, 1, flags=ra.FLAG_ACCUMULATE, chunk_spp=2)
    assert (bits(ctx.read_accum()) == bits(whole)).all()
    assert np.isfinite(whole).all() and whole.min() >= 0.0 and whole.max() <= spp
    fb = ctx.resolve(spp)
    top = ((fb.reshape(rows, cols)[:40] >> 8) & 0xFF).mean()           # sky rows: blue channel near 255
    assert top > 230 and (fb & 0xFF == 0).all()


def test_bench_workload_properties(ra, ctx, golden_scenes):
    """The bench workload itself (BASELINE configuration 2: 1200x800, 500 spp, depth 50): exact counters, bit-for-bit
    repeatability, additivity over a split of the sample range (two chunks of 250 = two accumulated launches of 250), and
    agreement of the automatic chunking (64 chunks, another summation order) to float accuracy."""
    cols, rows, spp = 1200, 800, 500
    ctx.set_filter(ra.FILTER_MFMA)                                       # bench.py's headline path: the linear list
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(cols, rows))
    _, st = ctx.render(cols, rows, 0, spp, 50, 1, chunk_spp=250)
    whole = ctx.read_accum()
    assert st["samples"] == cols * rows * spp and st["filter_misses"] == 0
    assert st["term_miss"] + st["term_absorb"] + st["term_depth"] == st["samples"]
    assert st["sphere_tests"] == st["rays"] * 485 and 2.7 < st["rays"] / st["samples"] < 2.8
    assert st["candidates"] < 2.2 * st["rays"]                           # the filter passes about two spheres per ray on
    _, st2 = ctx.render(cols, rows, 0, spp, 50, 1, chunk_spp=250)
    assert (bits(ctx.read_accum()) == bits(whole)).all() and st2["rays"] == st["rays"]
    ctx.render(cols, rows, 0, 250, 50, 1, chunk_spp=250, want_stats=False)
    ctx.render(cols, rows, 250, 250, 50, 1, flags=ra.FLAG_ACCUMULATE, chunk_spp=250, want_stats=False)
    assert (bits(ctx.read_accum()) == bits(whole)).all()
    _, st3 = ctx.render(cols, rows, 0, spp, 50, 1)                       # what bench.py launches
    auto = ctx.read_accum()
    assert st3["rays"] == st["rays"] and st3["term_depth"] == st["term_depth"]
    np.testing.assert_allclose(auto, whole, rtol=2e-5, atol=1e-4)
    assert np.isfinite(whole).all() and whole.min() >= 0.0 and whole.max() <= spp
    fb_a, fb_w = ctx.resolve(spp), None
    ctx.write_accum(cols, rows, whole)
    fb_w = ctx.resolve(spp)
    assert int(((fb_a ^ fb_w) != 0).sum()) < 100                         # 8-bit images agree but for a few rounding ties


def test_config5_full_size_properties(ra, ctx, golden_scenes, monkeypatch):
    """BASELINE configuration 5 at full size (20 164 spheres, 1200x800, 100 spp, depth 50): exact counters, and the streamed
    table (default at this size) against per-wave reads of the table and against the VALU filter, bit for bit."""
    cols, rows, spp = 1200, 800, 100
    table = golden_scenes["scaled"]
    ctx.set_filter(ra.FILTER_MFMA)                                       # the linear list that the configuration names (the default would walk the grid)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    _, st = ctx.render(cols, rows, 0, spp, 50, 1, chunk_spp=4)
    streamed = ctx.read_accum()
    assert st["samples"] == cols * rows * spp and st["sphere_tests"] == st["rays"] * len(table)
    assert st["term_miss"] + st["term_absorb"] + st["term_depth"] == st["samples"]
    assert st["candidates"] < 3.0 * st["rays"]
    ctx.set_tuning("stream", 0)
    _, st2 = ctx.render(cols, rows, 0, spp, 50, 1, chunk_spp=4)
    assert (bits(ctx.read_accum()) == bits(streamed)).all() and st2["rays"] == st["rays"]
    ctx.set_tuning("stream", -1)
    ctx.set_filter(ra.FILTER_VALU)
    _, st3 = ctx.render(cols, rows, 0, 4, 50, 1, chunk_spp=4)            # the VALU filter is 6x slower here: one chunk of it
    valu = ctx.read_accum()
    ctx.set_filter(ra.FILTER_MFMA)
    ctx.render(cols, rows, 0, 4, 50, 1, chunk_spp=4, want_stats=False)
    assert (bits(ctx.read_accum()) == bits(valu)).all()


@pytest.mark.parametrize("scene,cols,rows,spp", [("book1", 3840, 2160, 32), ("glass", 1200, 800, 100)])
def test_configs_3_and_4_full_frame(ra, ctx, golden_scenes, scene, cols, rows, spp):
    """BASELINE configurations 3 (4K frame; 32 of its 4096 spp -- tools/c3_full.py runs them all) and 4 (glass-heavy, 100
    spp): exact counters, repeatability, and additivity over a split of the sample range."""
    ctx.upload_scene(golden_scenes[scene])
    ctx.set_camera(ra.book_camera(cols, rows))
    half = spp // 2
    _, st = ctx.render(cols, rows, 0, spp, 50, 1, chunk_spp=half)
    whole = ctx.read_accum()
    assert st["samples"] == cols * rows * spp and st["sphere_tests"] == st["rays"] * len(golden_scenes[scene])
    assert st["term_miss"] + st["term_absorb"] + st["term_depth"] == st["samples"]
    assert (2.6 < st["rays"] / st["samples"] < 2.8) if scene == "book1" else (2.9 < st["rays"] / st["samples"] < 3.2)
    ctx.render(cols, rows, 0, half, 50, 1, chunk_spp=half, want_stats=False)
    ctx.render(cols, rows, half, half, 50, 1, flags=ra.FLAG_ACCUMULATE, chunk_spp=half, want_stats=False)
    assert (bits(ctx.read_accum()) == bits(whole)).all()
    assert np.isfinite(whole).all() and whole.min() >= 0.0 and whole.max() <= spp


@pytest.mark.parametrize("torch_first", [False, True])
def test_one_hip_runtime_whichever_import_comes_first(torch_first):
    """torch bundles its own libamdhip64; librtx must share it, or a process that creates a context and THEN imports
    torch ends with two runtimes and torch sees no GPU (api._share_torch_hip_runtime)."""
    import subprocess
    import sys
    from conftest import ROOT
    use_rtx = "import raytracing_amd as ra; c = ra.Context(0); c.close()"
    use_torch = "import torch; assert float(torch.ones(4, device='cuda:0').sum().item()) == 4.0"
    code = "; ".join([use_torch, use_rtx] if torch_first else [use_rtx, use_torch])
    code += "; m = open('/proc/self/maps').read(); n = len({l.split()[-1] for l in m.splitlines() if 'libamdhip64' in l}); assert n == 1, n"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("cols,rows,spp,begin,knobs", [(96, 64, 37, 0, {}), (75, 50, 16, 3, {"max_chunks": 3}), (300, 200, 9, 0, {"items_per_lane": 4, "tail_factor": 1})])
def test_automatic_chunk_plan_is_bit_exact_against_the_oracle(ra, ctx, oracle, golden_scenes, cols, rows, spp, begin, knobs):
    """chunk_spp = 0 (what bench.py and the CLI launch): the library reports the chunk boundaries it chose
    (rtx_last_chunk_plan) and the oracle, summing in that same order, gives the same bits."""
    table = golden_scenes["book1"]
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    for k, v in knobs.items():
        ctx.set_tuning(k, v)
    primary, st = ctx.render(cols, rows, begin, spp, 50, 11, want_primary=True)
    accum = ctx.read_accum()
    chunk_spp, starts = ctx.last_chunk_plan()
    assert chunk_spp == 0 and starts[0] == 0 and starts[-1] == spp and (np.diff(starts.astype(np.int64)) > 0).all()
    o_accum, o_primary, o_st = oracle.render(table, oracle.book_camera(cols, rows), rows, cols, begin, spp, 50, 11, want_primary=True, chunk_starts=starts)
    assert (primary == o_primary).all() and (bits(accum) == bits(o_accum)).all()
    assert st["rays"] == o_st["rays"] and st["samples"] == o_st["samples"]
    ctx.render(cols, rows, begin, spp, 50, 11, chunk_spp=5)
    assert ctx.last_chunk_plan() == (5, None)
    with pytest.raises(ra.RtxError):
        ctx.set_tuning("no_such_knob", 1)
    with pytest.raises(ra.RtxError):
        ctx.set_tuning("block", 100)


def test_a_bad_environment_value_is_reported_not_swallowed():
    """RTX_* variables are read once, at rtx_create: a value that rtx_set_tuning would refuse makes the create fail with a message
    naming the variable (round 2 ignored it silently)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = "import raytracing_amd as ra\ntry:\n    ra.Context(0)\nexcept ra.RtxError as e:\n    print('REFUSED', e)\n"
    for var, val in (("RTX_BLOCK", "100"), ("RTX_TAIL_FACTOR", "0"), ("RTX_MAX_CHUNKS", "many"), ("RTX_FILTER", "bvh")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(os.environ, **{var: val}))
        assert r.returncode == 0 and "REFUSED" in r.stdout and var in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code + "print('OK')"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(os.environ, RTX_BLOCK="256", RTX_FILTER="grid"))
    assert r.returncode == 0 and "REFUSED" not in r.stdout and "OK" in r.stdout, r.stdout + r.stderr


def test_config3_at_its_own_sample_count(ra, ctx, golden_scenes):
    """BASELINE configuration 3 on one GPU at full size: 3840x2160, 4096 spp, depth 50 in ONE launch (about 4 s; its 2 / 4 / 8-GPU
    split traces the same samples, 2048 / 1024 / 512 per rank).  Counters add up, the partial sums stay under the 1 GiB cap, and
    the 8-bit image equals the image of two accumulated 2048-sample launches -- the split a two-GPU group makes -- within one
    level (the float sums are added in another order)."""
    cols, rows, spp = 3840, 2160, 4096
    npix = cols * rows
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(cols, rows))
    _, st = ctx.render(cols, rows, 0, spp, 50, 1)
    chunk_spp, starts = ctx.last_chunk_plan()
    assert chunk_spp == 0 and starts[-1] == spp and (len(starts) - 1) * npix * 16 <= 1 << 30
    assert st["samples"] == npix * spp and st["sphere_tests"] == st["rays"] * 485
    assert st["term_miss"] + st["term_absorb"] + st["term_depth"] == st["samples"]
    assert 2.6 < st["rays"] / st["samples"] < 2.8
    assert st["samples"] / st["kernel_ms"] / 1e3 > 6000.0                # Msamples/s: no tail, no fallback (8 400 measured)
    whole = ctx.read_accum()
    fb_whole = ctx.resolve(spp)
    assert np.isfinite(whole).all() and whole.min() >= 0.0 and whole.max() <= spp
    _, st_a = ctx.render(cols, rows, 0, spp // 2, 50, 1)
    _, st_b = ctx.render(cols, rows, spp // 2, spp // 2, 50, 1, flags=ra.FLAG_ACCUMULATE)
    assert st_a["rays"] + st_b["rays"] == st["rays"] and st_a["term_depth"] + st_b["term_depth"] == st["term_depth"]
    fb_split = ctx.resolve(spp)
    diff = np.zeros(npix, dtype=np.int64)
    for sh in (24, 16, 8):
        diff = np.maximum(diff, np.abs(((fb_whole >> sh) & 0xFF).astype(np.int64) - ((fb_split >> sh) & 0xFF).astype(np.int64)))
    assert diff.max() <= 1 and (diff != 0).mean() < 0.01
    np.testing.assert_allclose(ctx.read_accum(), whole, rtol=3e-5, atol=1e-3)


def test_resolve_into_pinned_host_memory(ra, ctx, golden_scenes):
    """rtx_host_alloc: a framebuffer in pinned, GPU-mapped host memory is written by the resolve kernel itself; the pixels equal
    those that arrive through the staging copy, also at an offset inside the block and with -d borders."""
    cols, rows, spp = 200, 120, 5
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(cols, rows))
    ctx.render(cols, rows, 0, spp, 50, 3, want_stats=False)
    want, want_d = ctx.resolve(spp), ctx.resolve(spp, 16)
    block = ctx.host_framebuffer(cols * rows + 64)
    block[:] = 0xDEADBEEF
    got = ctx.resolve(spp, out=block[:cols * rows])
    assert (got == want).all() and (block[cols * rows:] == 0xDEADBEEF).all()
    block[:] = 0
    ctx.resolve(spp, 16, out=block[64:])
    assert (block[64:] == want_d).all() and not block[:64].any()
    small = ctx.host_framebuffer(100)
    with pytest.raises(ra.RtxError):
        ctx.host_free(np.zeros(4, dtype=np.uint32))                      # not ours
    ctx.host_free(small)
    ctx.host_free(block)
    assert ra.lib().rtx_host_alloc(ctx._h, 0, None) == -1               # RTX_ERR_ARG


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_root_quotient_identity_on_the_device(ra, mode):
    """The grid variant divides its roots by a = d.d as (float)((double)x * (1.0 / (double)a)) (trace_kernel.hip, `quot`): one double
    reciprocal per ray instead of an IEEE division per root.  The identity with x / a is argued in the kernel's comment; here the
    device checks it on 2^32 operand pairs per mode -- arbitrary bit patterns, the kernel's own range, operands with few significant
    bits (quotients near rounding boundaries) -- counting every pair whose two quotients differ in a bit (NaN on both sides and
    subnormal results on both sides aside: a root is only compared with 0.001 and FLT_MAX).  Mode 3: the kernel's 9-instruction
    square root (sqrt_rn) against the compiler's sqrtf on every one of the 2^32 bit patterns."""
    import ctypes as C
    L = ra.lib()
    L.rtx_debug_quotient_check.restype = C.c_int
    L.rtx_debug_quotient_check.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    ctx = ra.Context(0)
    pairs, bad = C.c_ulonglong(0), C.c_ulonglong(0)
    assert L.rtx_debug_quotient_check(ctx._h, 20260 + mode, mode, 65536, 256, C.byref(pairs), C.byref(bad)) == 0
    ctx.close()
    assert pairs.value == 1 << 32 and bad.value == 0, (pairs.value, bad.value)
