"""Samples-per-pixel split across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The path shards by sample index: rank r of R traces the global samples [begin_r, begin_r + count_r) of EVERY pixel
into its own full-frame float accumulation buffer.  The Philox counter carries the global sample index, so the union
over ranks is exactly the 1-GPU sample set and the primary-hit maps do not depend on the GPU count.  The one exchange
step is a sum of the accumulation buffers (all-reduce, rows*cols*3 floats: 11.5 MB at 1200x800), after which any rank
can resolve to 8 bits.  The reference has no counterpart (it is single-process).
"""
import numpy as np


def spp_range(rank, world, spp_begin, spp_count):
    """Contiguous share of [spp_begin, spp_begin+spp_count) for `rank`; remainder samples go to the lowest ranks."""
    base, rem = divmod(spp_count, world)
    count = base + (1 if rank < rem else 0)
    begin = spp_begin + rank * base + min(rank, rem)
    return begin, count


def render_split(render_fn, allreduce_fn, rank, world, spp_begin, spp_count):
    """render_fn(begin, count) -> accumulation buffer of this rank's samples (None / no call when count == 0 is not
    allowed: every rank must own a buffer, a rank without samples contributes zeros).  allreduce_fn(buf) sums in place."""
    begin, count = spp_range(rank, world, spp_begin, spp_count)
    buf = render_fn(begin, count)
    if world > 1:
        allreduce_fn(buf)
    return buf


class TorchAccum:
    """A torch-owned accumulation buffer bound to a Context (rtx_use_accum_buffer), summed across ranks with
    torch.distributed.all_reduce -- RCCL when the process group's backend is "nccl"."""

    def __init__(self, ctx, cols, rows, device):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.t = torch.zeros(rows * cols * 3, dtype=torch.float32, device=device)
        ctx.use_accum_buffer(cols, rows, self.t.data_ptr())

    def allreduce(self):
        import torch.distributed as dist
        self.ctx.sync()                       # the trace ran on the context's own stream
        dist.all_reduce(self.t, op=dist.ReduceOp.SUM)
        self.torch.cuda.synchronize()

    def release(self):
        self.ctx.use_accum_buffer(0, 0, 0)
