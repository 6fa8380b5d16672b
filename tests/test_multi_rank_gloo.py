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
