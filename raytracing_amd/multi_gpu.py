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
    """Contiguous share of [spp_begin, spp_begin+spp_count) for `rank`; remainder samples go to the lowest ranks.
    (The same rule as librtx's rtx_spp_range, which rtx_group_render splits by; tests/test_host_cabi.py compares them.)"""
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


def pass_ranges(spp_begin, spp_total, pass_spp):
    """The passes of a progressive render: [(begin, count)] covering [spp_begin, spp_total) in steps of pass_spp
    (0: one pass)."""
    step = pass_spp if pass_spp and pass_spp > 0 else max(spp_total - spp_begin, 1)
    return [(b, min(step, spp_total - b)) for b in range(spp_begin, spp_total, step)]


def render_progressive(render_fn, snapshot_fn, allreduce_fn, rank, world, spp_total, pass_spp, spp_done=0, preview_every=1,
                       on_preview=None):
    """Progressive spp-split render with a periodic exchange.

    Every pass covers a range of GLOBAL sample indices, split over the ranks like a one-shot render (spp_range), so a
    preview after k passes is exactly the image of the first samples.  Each rank ADDS its share to its own buffer:
    render_fn(begin, count) accumulates (it is not called with count == 0).  After every `preview_every` passes and
    after the last one the ranks exchange: snapshot_fn() returns a COPY of the rank's buffer, allreduce_fn(copy) sums
    it in place -- the running buffers are never summed into, or later passes would count the other ranks' samples
    again -- and on_preview(copy, samples_done) runs on every rank (return False to stop; all ranks must agree).
    Returns (summed buffer of the last exchange, samples done).  `spp_done` > 0 resumes: the caller has put the
    checkpointed sum into rank 0's buffer and zeros into the others."""
    done, last = spp_done, None
    passes = pass_ranges(spp_done, spp_total, pass_spp)
    for k, (begin, count) in enumerate(passes):
        b, c = spp_range(rank, world, begin, count)
        if c:
            render_fn(b, c)
        done = begin + count
        if (k + 1) % max(preview_every, 1) == 0 or k + 1 == len(passes):
            last = snapshot_fn()
            if world > 1:
                allreduce_fn(last)
            if on_preview is not None and on_preview(last, done) is False:
                break
    if last is None:                      # nothing left to render (resumed a finished job): exchange what there is
        last = snapshot_fn()
        if world > 1:
            allreduce_fn(last)
        if on_preview is not None:
            on_preview(last, done)
    return last, done


class TorchAccum:
    """A torch-owned accumulation buffer bound to a Context (rtx_use_accum_buffer), summed across ranks with
    torch.distributed.all_reduce -- RCCL when the process group's backend is "nccl"."""

    def __init__(self, ctx, cols, rows, device):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.t = torch.zeros(rows * cols * 3, dtype=torch.float32, device=device)
        # librtx works on its own non-blocking stream, which never waits for torch's: anything torch has queued on a
        # buffer (here the zero fill) must be complete before its address is handed over
        torch.cuda.current_stream(device).synchronize()
        ctx.use_accum_buffer(cols, rows, self.t.data_ptr())

    def allreduce(self):
        import torch.distributed as dist
        self.ctx.sync()                       # the trace ran on the context's own stream
        dist.all_reduce(self.t, op=dist.ReduceOp.SUM)
        self.torch.cuda.synchronize()

    def snapshot(self):
        """A copy of this rank's running buffer (for an exchange that must not disturb it)."""
        self.ctx.sync()
        c = self.t.clone()
        self.torch.cuda.current_stream(self.t.device).synchronize()      # the copy runs on torch's stream; resolve_tensor reads it on librtx's
        return c

    def allreduce_tensor(self, t):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        self.torch.cuda.synchronize()

    def resolve_tensor(self, t, cols, rows, spp_total, debug_tile=0):
        """8-bit framebuffer of another buffer of the same size (a summed snapshot): the context resolves whatever
        buffer it is bound to, so bind `t`, resolve, bind the running buffer again."""
        self.ctx.use_accum_buffer(cols, rows, t.data_ptr())
        try:
            return self.ctx.resolve(spp_total, debug_tile)
        finally:
            self.ctx.use_accum_buffer(cols, rows, self.t.data_ptr())

    def release(self):
        self.ctx.use_accum_buffer(0, 0, 0)
