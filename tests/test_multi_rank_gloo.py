"""The N > 1 path on CPU: two processes, gloo, each tracing its share of the samples (the oracle stands in for the GPU
renderer here -- tests may use it), one all-reduce of the accumulation buffers, and the result must equal the 1-rank
render with the same chunking bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT

COLS, ROWS, SPP, DEPTH, SEED = 24, 16, 6, 50, 9


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from raytracing_amd.multi_gpu import render_split
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = np.load(os.path.join(GOLDEN, "scenes.npz"))["book1"]
    cam = pyoracle.book_camera(COLS, ROWS)

    def render_fn(begin, count):
        if count == 0:
            return torch.zeros(ROWS * COLS * 3, dtype=torch.float32)
        accum, _, _ = pyoracle.render(table, cam, ROWS, COLS, begin, count, DEPTH, SEED, nthreads=2)
        return torch.from_numpy(accum)

    buf = render_split(render_fn, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), rank, world, 0, SPP)
    if rank == 0:
        np.save(out_path, buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank(tmp_path, oracle, golden_scenes):
    out = str(tmp_path / "accum.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    # 1 rank, chunks of SPP/2: (0 + A0) + A1 == A0 + A1, the all-reduce's sum
    want, _, _ = oracle.render(golden_scenes["book1"], oracle.book_camera(COLS, ROWS), ROWS, COLS, 0, SPP, DEPTH, SEED, chunk_spp=SPP // 2)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    fb = oracle.resolve(got, ROWS, COLS, SPP)
    assert fb.shape == (ROWS * COLS,) and (fb & 0xFF == 0).all()


# ---- progressive render with a periodic exchange (raytracing_amd.multi_gpu.render_progressive) ----------------------

P_SPP, P_PASS = 10, 4          # passes (0,4) (4,4) (8,2): the last one is ragged and splits 1 + 1


def _progressive_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from raytracing_amd.multi_gpu import render_progressive
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = np.load(os.path.join(GOLDEN, "scenes.npz"))["book1"]
    cam = pyoracle.book_camera(COLS, ROWS)
    local = torch.zeros(ROWS * COLS * 3, dtype=torch.float32)
    ranges = []

    def render_fn(begin, count):
        assert count > 0
        ranges.append((begin, count))
        accum, _, _ = pyoracle.render(table, cam, ROWS, COLS, begin, count, DEPTH, SEED, nthreads=2)
        local.add_(torch.from_numpy(accum))

    def on_preview(t, done):
        if rank == 0:
            np.save(os.path.join(out_dir, "preview_%d.npy" % done), t.numpy())
        return True

    last, done = render_progressive(render_fn, local.clone, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), rank, world,
                                    P_SPP, P_PASS, on_preview=on_preview)
    assert done == P_SPP
    np.save(os.path.join(out_dir, "ranges_%d.npy" % rank), np.array(ranges, dtype=np.int64))
    np.save(os.path.join(out_dir, "local_%d.npy" % rank), local.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "final.npy"), last.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_progressive_two_ranks(tmp_path, oracle, golden_scenes):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_progressive_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    cam = oracle.book_camera(COLS, ROWS)
    # every global sample is traced exactly once, by the rank a one-shot split of its pass gives it to
    r0, r1 = np.load(tmp_path / "ranges_0.npy"), np.load(tmp_path / "ranges_1.npy")
    assert r0.tolist() == [[0, 2], [4, 2], [8, 1]] and r1.tolist() == [[2, 2], [6, 2], [9, 1]]
    # the running buffers were never summed into: each equals the rank's own passes added in order
    want_local = []
    for rr in (r0, r1):
        acc = np.zeros(ROWS * COLS * 3, dtype=np.float32)
        for b, c in rr:
            acc = acc + oracle.render(golden_scenes["book1"], cam, ROWS, COLS, int(b), int(c), DEPTH, SEED)[0]
        want_local.append(acc)
    for r in (0, 1):
        assert (np.load(tmp_path / ("local_%d.npy" % r)).view(np.uint32) == want_local[r].view(np.uint32)).all()
    final = np.load(tmp_path / "final.npy")
    assert (final.view(np.uint32) == (want_local[0] + want_local[1]).view(np.uint32)).all()
    assert (np.load(tmp_path / "preview_10.npy").view(np.uint32) == final.view(np.uint32)).all()
    # a preview after k passes is the image of the first samples (the order of the float sums differs from a one-shot render)
    for done in (4, 8, 10):
        one_shot = oracle.render(golden_scenes["book1"], cam, ROWS, COLS, 0, done, DEPTH, SEED)[0]
        np.testing.assert_allclose(np.load(tmp_path / ("preview_%d.npy" % done)), one_shot, rtol=2e-6, atol=1e-6)


def test_progressive_resume_of_a_finished_job_still_exchanges():
    from raytracing_amd.multi_gpu import pass_ranges, render_progressive
    assert pass_ranges(0, 10, 4) == [(0, 4), (4, 4), (8, 2)] and pass_ranges(8, 10, 4) == [(8, 2)] and pass_ranges(0, 7, 0) == [(0, 7)]
    seen = []
    last, done = render_progressive(lambda b, c: seen.append(("render", b, c)), lambda: "snap", None, 0, 1, 10, 4, spp_done=10,
                                    on_preview=lambda t, d: seen.append(("preview", t, d)))
    assert (last, done) == ("snap", 10) and seen == [("preview", "snap", 10)]
    # preview_every = 2: exchanges after passes 2 and 3 (the last), an early stop ends the loop
    seen.clear()
    render_progressive(lambda b, c: seen.append((b, c)), lambda: "s", None, 0, 1, 10, 4, preview_every=2, on_preview=lambda t, d: seen.append(d))
    assert seen == [(0, 4), (4, 4), 8, (8, 2), 10]
    seen.clear()
    _, done = render_progressive(lambda b, c: seen.append((b, c)), lambda: "s", None, 0, 1, 10, 4, on_preview=lambda t, d: False)
    assert done == 4 and seen == [(0, 4)]
