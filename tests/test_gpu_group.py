"""The multi-device C ABI (rtx_group_*): on the one GPU of a test box a group of one device must be rtx_render bit for
bit, with and without RCCL in the loop (RTX_GROUP_RCCL=1 builds a communicator of one rank, so ncclCommInitAll and
ncclAllReduce really execute); the C++ wrapper and rtx_cli -G go through the same calls.  Groups of more than one device
need a multi-GPU node: the sample split they use is checked on the host (tests/test_host_cabi.py) and the sum across
ranks by the 2-rank gloo tests; groups of 2-4 MEMBERS run here in loopback on the one device (RTX_GROUP_LOOPBACK=1, no RCCL);
N > 1 DEVICES is UNMEASURED on hardware until the driver's scaling run.  Needs an MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import raytracing_amd
    raytracing_amd.lib()
    return raytracing_amd


def _single(ra, table, cols, rows, spp, seed, passes):
    ctx = ra.Context(0)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    done, st = 0, None
    for count in passes:
        _, st = ctx.render(cols, rows, done, count, 50, seed, flags=ra.FLAG_ACCUMULATE if done else 0)
        done += count
    out = (ctx.read_accum(), ctx.resolve(spp), st)
    ctx.close()
    return out


def test_group_of_one_is_rtx_render_bit_for_bit(ra, golden_scenes):
    table = golden_scenes["book1"]
    cols, rows, spp, seed = 160, 100, 12, 4
    want_accum, want_fb, want_st = _single(ra, table, cols, rows, spp, seed, [spp])
    g = ra.Group([0])
    assert g.size() == 1
    g.upload_scene(table)
    g.set_camera(ra.book_camera(cols, rows))
    st = g.render(cols, rows, 0, spp, 50, seed)
    g.allreduce()
    fb = g.resolve(spp)
    accum = g.member(0).read_accum()
    assert (bits(accum) == bits(want_accum)).all() and (fb == want_fb).all()
    for k in ("samples", "rays", "sphere_tests", "term_miss", "term_absorb", "term_depth", "candidates"):
        assert st[k] == want_st[k], k
    # progressive: two passes with an exchange in between count every sample once
    want2, _, _ = _single(ra, table, cols, rows, spp, seed, [5, 7])
    g.render(cols, rows, 0, 5, 50, seed, want_stats=False)
    g.allreduce()
    g.render(cols, rows, 5, 7, 50, seed, flags=ra.FLAG_ACCUMULATE, want_stats=False)
    g.allreduce()
    assert (bits(g.member(0).read_accum()) == bits(want2)).all()
    g.close()
    with pytest.raises(ra.RtxError, match="twice"):
        ra.Group([0, 0])
    with pytest.raises(ra.RtxError):
        ra.Group([0, 99])
    with pytest.raises(ra.RtxError):
        ra.Group([])


def test_group_exchange_through_rccl_on_one_rank(golden_scenes):
    """A fresh process with RTX_GROUP_RCCL=1: ncclCommInitAll over one device and an in-place ncclAllReduce(sum, float32) on
    the member's stream -- the collective's call chain on hardware; the buffer must come back unchanged."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import raytracing_amd as ra
table = np.load(%r)["book1"]
cols, rows, spp = 120, 80, 6
ctx = ra.Context(0); ctx.upload_scene(table); ctx.set_camera(ra.book_camera(cols, rows))
ctx.render(cols, rows, 0, spp, 50, 3); want = ctx.read_accum(); ctx.close()
assert "rccl" not in open("/proc/self/maps").read()        # rendering on one GPU has not loaded RCCL
g = ra.Group([0]); g.upload_scene(table); g.set_camera(ra.book_camera(cols, rows))
g.render(cols, rows, 0, spp, 50, 3); g.allreduce(); g.allreduce()
got = g.member(0).read_accum()
assert (got.view(np.uint32) == want.view(np.uint32)).all()
assert "rccl" in open("/proc/self/maps").read()
g.close(); print("rccl exchange ok")
''' % (ROOT, os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    env = dict(os.environ, RTX_GROUP_RCCL="1", NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl exchange ok" in r.stdout, r.stdout + r.stderr


def test_cli_gpu_count_flag(tmp_path, ra, oracle, golden_scenes):
    """rtx_cli -G 1 -D 0 renders through rtx_group_*; the file equals the oracle's pixels.  -G 0 is refused."""
    cli = os.path.join(ROOT, "raytracing_amd", "rtx_cli")
    out = tmp_path / "g.ppm"
    r = subprocess.run([cli, "-g", "-G", "1", "-D", "0", "-w", "96", "-h", "64", "-a", "3", "-m", "50", "-s", "1", "-f", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    px = np.loadtxt(str(out), skiprows=3, dtype=np.uint32).reshape(-1, 3)
    ctx = ra.Context(0)
    ctx.upload_scene(golden_scenes["book1"])
    ctx.set_camera(ra.book_camera(96, 64))
    ctx.render(96, 64, 0, 3, 50, 1, want_stats=False)
    fb = ctx.resolve(3)
    ctx.close()
    assert ((px[:, 0] << 24 | px[:, 1] << 16 | px[:, 2] << 8) == fb).all()
    r = subprocess.run([cli, "-g", "-G", "0", "-f", str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0


def test_groups_of_two_and_three_members_in_loopback(golden_scenes):
    """RTX_GROUP_LOOPBACK=1: groups of SEVERAL members on the one GPU of a test box (the exchange is a plain sum into member 0,
    everything else is the multi-GPU code): the sample split, the per-member launches, the sum, the zeroing of the other
    members, progressive passes and the statistics.  With the members' ranges as the chunks of a single launch the sums are
    added in the same order, so the group's image equals rtx_render's bit for bit."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import raytracing_amd as ra
table = np.load(%r)["book1"]
cols, rows, seed = 120, 80, 3
cam = ra.book_camera(cols, rows)
for world, spp in ((2, 8), (3, 9), (3, 8), (4, 2)):
    per = [ra.spp_range(r, world, 0, spp) for r in range(world)]
    g = ra.Group([0] * world); g.upload_scene(table); g.set_camera(cam)
    st = g.render(cols, rows, 0, spp, 50, seed, chunk_spp=spp)          # every member: its share as one chunk
    g.allreduce()
    got, fb = g.member(0).read_accum(), g.resolve(spp)
    for r in range(1, world):
        assert not g.member(r).read_accum().any()                       # the invariant: the sum over the members is the image
    # the same sums in the same order on one context: one launch per share, accumulated
    ctx = ra.Context(0); ctx.upload_scene(table); ctx.set_camera(cam)
    tot = dict(samples=0, rays=0)
    for k, (b, c) in enumerate(per):
        if c == 0:
            continue
        _, s1 = ctx.render(cols, rows, b, c, 50, seed, flags=ra.FLAG_ACCUMULATE if k else 0, chunk_spp=c)
        tot["samples"] += s1["samples"]; tot["rays"] += s1["rays"]
    want, want_fb = ctx.read_accum(), ctx.resolve(spp)
    assert (got.view(np.uint32) == want.view(np.uint32)).all() and (fb == want_fb).all(), (world, spp)
    assert st["samples"] == tot["samples"] == cols * rows * spp and st["rays"] == tot["rays"]
    # a second, accumulating pass after the exchange counts every sample once
    g.render(cols, rows, spp, 4, 50, seed, flags=ra.FLAG_ACCUMULATE, chunk_spp=4, want_stats=False)
    g.allreduce()
    for k, (b, c) in enumerate(ra.spp_range(r, world, spp, 4) for r in range(world)):
        if c:
            ctx.render(cols, rows, b, c, 50, seed, flags=ra.FLAG_ACCUMULATE, chunk_spp=c, want_stats=False)
    assert (g.member(0).read_accum().view(np.uint32) == ctx.read_accum().view(np.uint32)).all(), (world, "second pass")
    ctx.close(); g.close()
print("loopback groups ok")
''' % (ROOT, os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    env = dict(os.environ, RTX_GROUP_LOOPBACK="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "loopback groups ok" in r.stdout, r.stdout + r.stderr


def test_cli_with_three_members_in_loopback(tmp_path):
    """rtx_cli -G 3 under RTX_GROUP_LOOPBACK=1: the C++ wrapper's multi-member path (split, exchange per pass, checkpoint of the
    sum on member 0, resume with the other members zeroed) on the one GPU.  The image equals -G 1's up to the order of the
    float sums; a render resumed from its checkpoint equals the uninterrupted one exactly."""
    cli = os.path.join(ROOT, "raytracing_amd", "rtx_cli")
    env = dict(os.environ, RTX_GROUP_LOOPBACK="1")
    base = [cli, "-w", "96", "-h", "64", "-m", "8", "-s", "4", "-6"]

    def pixels(path):
        return np.frombuffer(open(path, "rb").read()[-96 * 64 * 3:], dtype=np.uint8).astype(int)

    one, three, part, ck = tmp_path / "one.ppm", tmp_path / "three.ppm", tmp_path / "part.ppm", tmp_path / "c.accum"
    r = subprocess.run(base + ["-G", "1", "-a", "12", "-f", str(one)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(base + ["-G", "3", "-a", "12", "-p", "6", "-f", str(three)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and r.stdout.count("pass done") == 2, r.stdout + r.stderr
    assert np.abs(pixels(one) - pixels(three)).max() <= 1 and (pixels(one) != pixels(three)).mean() < 0.01
    # stopped after one pass, then resumed by three members from the checkpoint
    r = subprocess.run(base + ["-G", "3", "-a", "6", "-p", "6", "-k", str(ck), "-f", str(part)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and ck.exists(), r.stdout + r.stderr
    r = subprocess.run(base + ["-G", "3", "-a", "12", "-p", "6", "-k", str(ck), "-R", "-f", str(part)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "resuming" in r.stdout and r.stdout.count("pass done") == 1, r.stdout + r.stderr
    assert open(part, "rb").read() == open(three, "rb").read()
    # without the rehearsal switch a second GPU is really asked for
    r = subprocess.run(base + ["-G", "3", "-a", "2", "-f", str(part)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
