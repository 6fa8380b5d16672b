"""CPU-side checks of the product's host layer: the C-ABI library loads and exports what include/rtx.h declares, and
its GPU-free entry points (camera, scene generator, PPM writer) agree with the reference-made fixtures.  No GPU work."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, bits


@pytest.fixture(scope="module")
def ra():
    import raytracing_amd
    raytracing_amd.lib()
    return raytracing_amd


def test_library_exports_every_declared_symbol(ra):
    hdr = open(os.path.join(ROOT, "include", "rtx.h")).read()
    declared = set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    L = ra.lib()
    for name in declared:
        assert hasattr(L, name), name
    from raytracing_amd import api
    assert declared == set(api.EXPORTS)


def test_no_cpu_fallback(ra):
    """Without a GPU the backend must refuse loudly (this test is skipped on a GPU box)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ra.RtxError, match="no HIP device|no CPU path"):
        ra.Context(0)
    fb = np.zeros(4, dtype=np.uint32)
    with pytest.raises(ra.RtxError):
        ra.render_scene_hip(ra.Scene.from_table(ra.random_scene(1)), ra.book_camera(2, 2), 2, 2, fb, 1, 5)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracing_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("pyoracle", "librt_oracle", "rt_oracle.h", "oracle/_ref", "ref_harness"):
                    assert needle not in text, (f, needle)
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    assert "oracle" not in text


def test_scene_generator_matches_golden_and_oracle(ra, oracle, golden_scenes):
    import json
    with open(os.path.join(GOLDEN, "scenes.json")) as f:
        meta = json.load(f)
    for key, kw in meta.items():
        t = ra.random_scene(kw["seed"], kw["grid_half"], kw["p_diffuse"], kw["p_metal"])
        assert t.shape == (kw["n"], 10)
        assert (bits(t) == bits(golden_scenes[key])).all()
    a = ra.random_scene(7)
    b = oracle.random_scene(7)
    assert (bits(a) == bits(b)).all()
    c = ra.random_scene(8)
    k = min(len(a), len(c))
    assert not (bits(a[:k, :3]) == bits(c[:k, :3])).all()   # the seed matters
    # layout of RayTracing.cpp:186 and :205-207: ground first, the three big spheres last
    assert tuple(a[0, :5]) == (0.0, -1000.0, 0.0, 1000.0, 1.0)
    assert [tuple(r[:5]) for r in a[-3:]] == [(-4.0, 1.0, 0.0, 1.0, 1.0), (0.0, 1.0, 0.0, 1.0, 3.0), (4.0, 1.0, 0.0, 1.0, 2.0)]


def test_camera_matches_reference_constructor(ra, golden_kat):
    for r in golden_kat["kat_camera"]:
        cam = ra.Camera(r[0], r[1], r[2], r[3], r[4:7], r[7:10], r[10:13])
        assert (bits(cam.as_array()) == bits(r[13:35])).all()


def test_scene_mirror_roundtrip(ra):
    t = ra.random_scene(3)
    s = ra.Scene.from_table(t)
    assert len(s.objects) == len(t) and (bits(s.table()) == bits(t)).all()
    m = ra.Material()                                       # material.h:23-30 defaults
    assert (m.type, m.albedo, m.blur, m.refractionIndex) == (0, (1.0, 0.0, 1.0), 1.0, 1.0)
    s.objects.append(ra.Sphere((0, 0, 0), 1.0, None))
    with pytest.raises(ra.RtxError):
        s.table()


def test_ppm_writer_format(ra, tmp_path):
    fb = np.array([0x01020300, 0xFFFEFD00, 0x00000000, 0x80402000, 0x0A0B0CFF, 0x7F7F7F00], dtype=np.uint32)
    p = tmp_path / "out.ppm"
    ra.write_ppm(p, 3, 2, fb)
    lines = open(p).read().split("\n")
    assert lines[:3] == ["P3", "3 2", "255"]                 # RayTracing.cpp:152-154
    assert lines[3:9] == ["1 2 3", "255 254 253", "0 0 0", "128 64 32", "10 11 12", "127 127 127"]
    assert lines[9] == ""


def test_ppm_p6_writer(ra, tmp_path):
    fb = np.array([0x01020300, 0xFFFEFD00, 0x00000000, 0x80402000, 0x0A0B0CFF, 0x7F7F7F00], dtype=np.uint32)
    p = tmp_path / "out6.ppm"
    ra.write_ppm(p, 3, 2, fb, binary=True)
    raw = open(p, "rb").read()
    assert raw.startswith(b"P6\n3 2\n255\n")
    assert raw[len(b"P6\n3 2\n255\n"):] == bytes([1, 2, 3, 255, 254, 253, 0, 0, 0, 128, 64, 32, 10, 11, 12, 127, 127, 127])


def test_spp_partition():
    from raytracing_amd.multi_gpu import spp_range
    for spp, world, begin in [(500, 8, 0), (4096, 8, 0), (10, 4, 3), (3, 8, 0), (1, 1, 9)]:
        parts = [spp_range(r, world, begin, spp) for r in range(world)]
        assert parts[0][0] == begin and sum(c for _, c in parts) == spp
        for (b0, c0), (b1, _) in zip(parts, parts[1:]):
            assert b0 + c0 == b1
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_dropin_demo_fails_loudly_without_a_gpu(tmp_path, oracle, golden_scenes):
    """The reference's own types through renderSceneHIP (oracle/_ref/dropin_demo): on a box without a GPU the call must
    fail with the library's message -- there is no CPU path to fall back to."""
    import subprocess
    from conftest import ROOT
    demo = os.path.join(ROOT, "oracle", "_ref", "dropin_demo")
    if not os.path.exists(demo):
        pytest.skip("oracle/_ref/dropin_demo is built only where the reference tree exists")
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: tests/test_gpu_dropin.py covers the working case")
    except ImportError:
        pass
    scene_bin = tmp_path / "scene.bin"
    oracle.write_scene_bin(str(scene_bin), golden_scenes["book1"])
    r = subprocess.run([demo, str(scene_bin), str(tmp_path / "o.fb"), "32", "16", "1", "3", "0"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no HIP device" in r.stdout + r.stderr


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/rtx.h is the boundary a cgo / JNI / ctypes stub binds: it must compile as strict C99 and the library must
    link from a C program (no C++ types, no torch types in the signatures)."""
    import subprocess
    src = tmp_path / "use_rtx.c"
    src.write_text(r'''
#include "rtx.h"
#include <stdio.h>
int main(void)
{
    rtx_camera cam;
    const float pos[3] = { 13.0f, 2.0f, 3.0f }, up[3] = { 0.0f, 1.0f, 0.0f }, lookat[3] = { 0.0f, 0.0f, 0.0f };
    rtx_ctx* ctx = 0;
    int rc = rtx_camera_init(&cam, 20.0f, 1.5f, 0.1f, 10.0f, pos, up, lookat);
    if (rc != RTX_OK) return 2;
    printf("%s lens %.3f cap %u\n", rtx_version(), (double)cam.lens_radius, rtx_scene_capacity(11));
    rc = rtx_create(0, &ctx);
    if (rc == RTX_OK) { rtx_destroy(ctx); printf("gpu\n"); return 0; }
    printf("no gpu: %s\n", rtx_last_error());
    return 0;
}
''')
    exe = tmp_path / "use_rtx"
    libdir = os.path.join(ROOT, "raytracing_amd")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-lrtx", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "raytracing_amd" in r.stdout and "lens 0.050" in r.stdout and ("gpu" in r.stdout)


def test_chunk_plan_properties():
    """rtx_plan_chunks (pure host code): the automatic sample-chunk plan covers the range exactly, shrinks monotonically to
    small items at the end, and stays inside its work-item budget -- 1200x800 at 500 spp under 700 MiB of partial sums
    (round 1: 983 MB), 3840x2160 at 4096 spp under 1 GiB (round 1: 8.5 GB)."""
    import raytracing_amd.api as api
    lanes = 256 * 1024            # MI355X: 256 CUs x 4 SIMDs x 4 resident waves of this kernel x 64 lanes
    for spp, npix, kw in [(500, 1200 * 800, {}), (4096, 3840 * 2160, {}), (100, 1200 * 800, {}), (10, 1200 * 800, {}), (1, 7, {}), (3, 1, {}),
                          (500, 1200 * 800, {"max_items": 4 << 20}), (500, 1200 * 800, {"max_items": 1200 * 800}), (8192, 120 * 80, {}),
                          (977, 300 * 200, {"max_chunks": 5}), (64, 1200 * 800, {"tail_factor": 2}), (50, 640 * 480, {"items_per_lane": 16})]:
        t = api.plan_chunks(spp, npix, lanes, **kw)
        d = np.diff(t.astype(np.int64))
        assert t[0] == 0 and t[-1] == spp and (d > 0).all(), (spp, npix, kw, t)
        assert (d[1:] <= d[:-1]).all(), (spp, npix, kw, d)                      # large chunks first
        assert len(d) <= min(63, kw.get("max_chunks", 63)) and len(d) * npix <= kw.get("max_items", 1 << 26)
    bench = api.plan_chunks(500, 1200 * 800, lanes)
    assert (len(bench) - 1) * 1200 * 800 * 16 < 700 << 20 and bench[-1] - bench[-2] == 2          # 2-sample items at the end, never 1 from 48 spp on
    assert min(np.diff(api.plan_chunks(24, 1200 * 800, lanes).astype(np.int64))) == 1
    big = api.plan_chunks(4096, 3840 * 2160, lanes)
    assert (len(big) - 1) * 3840 * 2160 * 16 < 1 << 30
    with pytest.raises(api.RtxError):
        api.plan_chunks(10, 100, lanes, max_items=99)


def test_spp_range_of_the_library_equals_the_python_hosts():
    """rtx_spp_range (what rtx_group_render splits by) and multi_gpu.spp_range (the one-process-per-GPU host) are one rule."""
    import raytracing_amd as ra
    from raytracing_amd.multi_gpu import spp_range
    for spp, world, begin in [(500, 8, 0), (4096, 8, 0), (10, 4, 3), (3, 8, 0), (1, 1, 9), (7, 3, 100), (500, 7, 11)]:
        lib_parts = [ra.spp_range(r, world, begin, spp) for r in range(world)]
        assert lib_parts == [spp_range(r, world, begin, spp) for r in range(world)]
        assert sum(c for _, c in lib_parts) == spp


def test_library_resolves_rccl_when_a_group_needs_it():
    """The multi-device exchange lives under the C ABI, but a process that renders on one GPU must not pay for RCCL: librtx.so has no
    link-time dependency on librccl and resolves the seven entry points it uses (dlsym) when the first group of several GPUs is made."""
    import subprocess
    from conftest import ROOT
    lib = os.path.join(ROOT, "raytracing_amd", "librtx.so")
    out = subprocess.check_output(["nm", "-D", lib], text=True)
    assert " U nccl" not in out
    needed = subprocess.check_output(["readelf", "-d", lib], text=True)
    assert "rccl" not in needed
    blob = open(lib, "rb").read()
    for sym in (b"ncclAllReduce", b"ncclReduce", b"ncclCommInitAll", b"ncclCommDestroy", b"ncclGroupStart", b"ncclGroupEnd", b"ncclGetErrorString"):
        assert sym + b"\0" in blob, sym


def _grid_cells_of(g, lo, hi):
    """Cell index ranges (inclusive, clamped) that the box [lo, hi] touches."""
    a = np.clip(np.floor((lo - g["gmin"]) / g["h"]), 0, g["dim"] - 1).astype(np.int64)
    b = np.clip(np.floor((hi - g["gmin"]) / g["h"]), 0, g["dim"] - 1).astype(np.int64)
    return a, b


def _blur(r, l):
    """How far beyond its surface the reference's fp32 discriminant can still accept a ray, origin l away (rtx_host.cpp)."""
    r = abs(float(r))
    return np.sqrt(r * r + (l * l + r * r) * 2.0 ** -19) - r


@pytest.mark.parametrize("density", [12, 35, 400])
def test_grid_construction_guarantees(ra, golden_scenes, density):
    """rtx_build_grid (the host half of RTX_FILTER_GRID): every sphere is either on the big list or, for each registration tier
    k, in parts 0..k of every cell that its bounding box grown by pad + blur(r, tier_l[k]) touches; parts ascend; the grid box
    contains every grown box; the ground sphere is big; offsets are consistent; the tiers' distance limits leave room for a
    cell, a sphere and the growth."""
    for name in ("book1", "scaled"):
        table = golden_scenes[name]
        g = ra.build_grid(table, density)
        dim, start, items, big, tend = g["dim"], g["cell_start"], g["items"], set(g["big"].tolist()), g["tier_end"].astype(np.int64)
        ncell = int(np.prod(dim))
        assert start[0] == 0 and start[-1] == len(items) and (np.diff(start.astype(np.int64)) >= 0).all()
        assert (tend[:, 0] >= start[:-1]).all() and (tend[:, 1] >= tend[:, 0]).all() and (tend[:, 2] >= tend[:, 1]).all() and (tend[:, 2] == start[1:]).all()
        assert (dim >= 1).all() and (dim <= 256).all() and g["pad"] > 0
        assert 0 in big and len(big) < 8                      # the ground sphere (index 0, radius 1000) and at most the three large ones
        ext = g["h"].astype(np.float64) * dim
        diag = np.sqrt(((ext - 2 * g["tier_grow"][2]) ** 2).sum())
        assert 1.1 * diag * (1 - 1e-3) <= g["tier_l"][2] <= 2 * diag * (1 + 1e-3)
        assert np.allclose(g["tier_l"][:2], [0.25 * g["tier_l"][2], 0.5 * g["tier_l"][2]], rtol=1e-9)
        cell_diag = np.sqrt((g["h"].astype(np.float64) ** 2).sum())
        r_max = max(abs(float(r)) for i, r in enumerate(table[:, 3]) if i not in big)
        for k in range(2):
            assert g["tier_s"][k] <= g["tier_l"][k] - (cell_diag + r_max + g["tier_grow"][k]) + 1e-6 or g["tier_s"][k] == 0
        assert g["tier_s"][2] <= g["tier_l"][2]                     # the limit for origins: beyond it the table is scanned
        upto = []                                                   # per tier: cell -> set of spheres in parts 0..k
        for k in range(3):
            upto.append([set(items[start[c]:tend[c, k]].tolist()) for c in range(ncell)])
        for c in range(ncell):
            prev = start[c]
            for k in range(3):
                part = items[prev:tend[c, k]].astype(np.int64)
                assert (np.diff(part) > 0).all()                  # ascending, no duplicates inside a part
                prev = tend[c, k]
            assert len(upto[2][c]) == tend[c, 2] - start[c]        # and none across the parts
        for i, row in enumerate(table):
            if i in big:
                assert all(i not in s for s in upto[2])
                continue
            ctr, r = row[0:3].astype(np.float64), abs(float(row[3]))
            for k in range(3):
                grow = 0.999 * g["pad"] + _blur(r, g["tier_l"][k])
                assert grow <= g["tier_grow"][k] * (1 + 1e-9)
                assert ((ctr - r - grow) >= g["gmin"] - 1e-4 * ext).all() and ((ctr + r + grow) <= g["gmin"] + ext * (1 + 1e-4)).all()
                a, b = _grid_cells_of(g, ctr - r - grow, ctr + r + grow)
                for z in range(a[2], b[2] + 1):
                    for y in range(a[1], b[1] + 1):
                        for x in range(a[0], b[0] + 1):
                            assert i in upto[k][(z * dim[1] + y) * dim[0] + x], (name, i, k, x, y, z)
        binned = len(table) - len(big)
        assert len(items) >= binned
        mean = binned / float(np.prod(dim))
        assert mean < 4.0 * density / 100.0 + 1.0              # resolution follows the density knob (thin axes collapse, hence the slack)


def test_grid_blur_bounds_the_fp32_discriminant():
    """The bound behind the registration tiers: over random (o, d, c, r) the fp32 discriminant of sphere.cpp:72-78, evaluated
    in the reference's operation order, differs from the real one by less than 20 u a (L^2 + r^2), u = 2^-24 -- so a ray
    that it accepts passes within blur(r, L) of the sphere."""
    rng = np.random.default_rng(5)
    f32 = np.float32
    worst = 0.0
    for _ in range(20000):
        scale = 10.0 ** rng.uniform(-2, 3)
        o = (rng.normal(size=3) * scale).astype(f32)
        c = (rng.normal(size=3) * scale).astype(f32)
        r = f32(abs(rng.normal()) * scale * 10.0 ** rng.uniform(-3, 0))
        d = rng.normal(size=3).astype(f32)
        oc = o - c
        bq = f32(f32(f32(oc[0] * d[0]) + f32(oc[1] * d[1])) + f32(oc[2] * d[2]))
        ococ = f32(f32(f32(oc[0] * oc[0]) + f32(oc[1] * oc[1])) + f32(oc[2] * oc[2]))
        a = f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))
        cq = f32(ococ - f32(r * r))
        disc = float(f32(f32(bq * bq) - f32(a * cq)))
        od, dd, cd, rd = o.astype(np.float64), d.astype(np.float64), c.astype(np.float64), float(r)
        ocd = od - cd
        true = float((ocd @ dd) ** 2 - (dd @ dd) * (ocd @ ocd - rd * rd))
        bound = 20.0 * 2.0 ** -24 * float(dd @ dd) * float(ocd @ ocd + rd * rd)
        worst = max(worst, abs(disc - true) / bound)
    assert worst < 1.0, worst


def test_grid_degenerate_inputs(ra):
    """No spheres, non-finite spheres and spheres that all exceed the grid give no cells and everything on the big list."""
    empty = ra.build_grid(np.zeros((0, 10), dtype=np.float32))
    assert empty["dim"].tolist() == [0, 0, 0] and len(empty["items"]) == 0 and len(empty["big"]) == 0
    t = np.zeros((3, 10), dtype=np.float32)
    t[:, 3] = 1.0
    t[0, 0] = np.nan
    t[1, 1] = np.inf
    g = ra.build_grid(t)
    assert set(g["big"].tolist()) >= {0, 1}
    assert all(i in set(g["big"].tolist()) or i in set(g["items"].tolist()) for i in range(3))
    one = ra.build_grid(np.array([[1, 2, 3, 0.5, 1, 0, 0, 0, 0, 1]], dtype=np.float32))
    assert (len(one["big"]) == 1) != (0 in set(one["items"].tolist()))
    assert ra.lib().rtx_build_grid(0, None, None, None, None, 0, None, None, 0, None, 0, None, 0, None, 0) != 0     # NULL info / density 0
