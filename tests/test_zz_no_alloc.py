"""rtx_render_noalloc -- the entry point that allocates nothing (static device scratch) -- against rtx_render, through the
C ABI, bit for bit: two calls in a row on one context, the second with a smaller input, then an accumulating pass, the
size limit and the refused flag.  Needs an MI355X.  Collected last (file name) so that a failure hides no other test."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _hip():
    """The HIP runtime librtx.so is bound to (already in the process once the library is loaded)."""
    import raytracing_amd
    raytracing_amd.lib()
    h = C.CDLL("libamdhip64.so.7")
    h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    h.hipFree.argtypes = [C.c_void_p]
    h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return h


def _free_bytes(hip):
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


class DeviceBuffer:
    """Caller-owned device memory: what a host that cannot let the library allocate would pass in."""

    def __init__(self, hip, n_floats):
        self.hip, self.n = hip, n_floats
        self.ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(self.ptr), n_floats * 4) == 0
        assert hip.hipMemset(self.ptr, 0xFF, n_floats * 4) == 0          # NaN pattern: every float must be overwritten

    def read(self):
        out = np.zeros(self.n, dtype=np.float32)
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, self.n * 4, 2) == 0      # hipMemcpyDeviceToHost
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def test_noalloc_matches_rtx_render_twice_in_a_row_then_accumulates(golden_scenes):
    import raytracing_amd as ra
    hip = _hip()
    table = golden_scenes["book1"]
    ctx = ra.Context(0)
    ctx.upload_scene(table)
    # the existing entry point first: its results are the expectation
    expect = []
    cases = [(300, 200, 0, 24, 0, 7), (96, 64, 3, 5, 2, 9)]             # (cols, rows, spp_begin, spp, chunk_spp, seed); the second is smaller
    for cols, rows, begin, spp, chunk, seed in cases:
        ctx.set_camera(ra.book_camera(cols, rows))
        ctx.render(cols, rows, begin, spp, 50, seed, chunk_spp=chunk, want_stats=False)
        expect.append((ctx.read_accum(), ctx.last_chunk_plan()))
    bufs = [DeviceBuffer(hip, c[0] * c[1] * 3) for c in cases]
    # two calls in a row through the no-alloc entry point, sharing the static scratch
    for (cols, rows, begin, spp, chunk, seed), buf, (want, plan) in zip(cases, bufs, expect):
        ctx.set_camera(ra.book_camera(cols, rows))
        ctx.render_noalloc(cols, rows, begin, spp, 50, seed, buf.ptr.value, chunk_spp=chunk)
        got_plan = ctx.last_chunk_plan()
        assert got_plan[0] == plan[0] and (plan[1] is None or (got_plan[1] == plan[1]).all())
        got = buf.read()
        assert np.isfinite(got).all() and (bits(got) == bits(want)).all()
    # RTX_FLAG_ACCUMULATE adds a second sample range to the caller's buffer, as rtx_render does to its own
    cols, rows, begin, spp, chunk, seed = cases[1]
    ctx.render(cols, rows, begin, spp, 50, seed, chunk_spp=chunk, want_stats=False)
    ctx.render(cols, rows, begin + spp, 4, 50, seed, flags=ra.FLAG_ACCUMULATE, chunk_spp=4, want_stats=False)
    want = ctx.read_accum()
    ctx.render_noalloc(cols, rows, begin + spp, 4, 50, seed, bufs[1].ptr.value, flags=ra.FLAG_ACCUMULATE, chunk_spp=4)
    assert (bits(bufs[1].read()) == bits(want)).all()
    # max_depth 0 never reaches the scratch (raytracer.cpp:250: every sample is (1,1,1))
    ctx.render_noalloc(cols, rows, 0, 3, 0, seed, bufs[1].ptr.value)
    assert (bufs[1].read() == 3.0).all()
    # limits: more pixels than the scratch has work items, a plan that does not fit, the flag that needs a stack, a host pointer
    with pytest.raises(ra.RtxError, match="work items"):
        ctx.render_noalloc(4096, 2048, 0, 1, 50, seed, bufs[0].ptr.value)
    with pytest.raises(ra.RtxError, match="work items"):
        ctx.render_noalloc(300, 200, 0, 200, 50, seed, bufs[0].ptr.value, chunk_spp=1)
    with pytest.raises(ra.RtxError, match="RECURSIVE"):
        ctx.render_noalloc(cols, rows, 0, 1, 50, seed, bufs[1].ptr.value, flags=ra.FLAG_RECURSIVE)
    host = np.zeros(cols * rows * 3, dtype=np.float32)
    with pytest.raises(ra.RtxError, match="device memory"):
        ctx.render_noalloc(cols, rows, 0, 1, 50, seed, host.ctypes.data)
    # a large frame whose automatic plan is cut down to the scratch: 1200 x 800 fits 4 chunks
    big = DeviceBuffer(hip, 1200 * 800 * 3)
    ctx.set_camera(ra.book_camera(1200, 800))
    ctx.render_noalloc(1200, 800, 0, 12, 50, seed, big.ptr.value)
    chunk_spp, starts = ctx.last_chunk_plan()
    assert chunk_spp == 0 and len(starts) - 1 <= 4 and starts[-1] == 12
    ctx.render(1200, 800, 0, 12, 50, seed, want_stats=False)
    np.testing.assert_allclose(big.read(), ctx.read_accum(), rtol=1e-5, atol=1e-5)        # same samples, another chunk plan
    # the property the entry point exists for: a call allocates nothing -- the device's free memory is the same before and
    # after, also for a frame size and a plan this context has not rendered before (rtx_render would grow its buffers here)
    odd = DeviceBuffer(hip, 640 * 360 * 3)
    ctx.set_camera(ra.book_camera(640, 360))
    before = _free_bytes(hip)
    ctx.render_noalloc(640, 360, 0, 9, 50, seed, odd.ptr.value)
    ctx.render_noalloc(1200, 800, 0, 5, 50, seed, big.ptr.value, chunk_spp=2)
    assert _free_bytes(hip) == before
    # a buffer too small for the image is refused, not overrun
    with pytest.raises(ra.RtxError, match="room"):
        ctx.render_noalloc(1200, 800, 0, 1, 50, seed, odd.ptr.value)
    for b in bufs + [big, odd]:
        b.free()
    ctx.close()


def test_overlapping_noalloc_calls_on_one_gpu_are_refused(golden_scenes):
    """Every call on a GPU shares that GPU's scratch array: two calls that overlap -- two contexts, two threads -- would race, so
    the second is refused with RTX_ERR_STATE while the first runs (and succeeds once it has ended)."""
    import threading
    import raytracing_amd as ra
    hip = _hip()
    cols, rows, spp = 1200, 800, 150                       # about 20 ms per call
    ctxs, bufs = [], []
    for _ in range(2):
        c = ra.Context(0)
        c.upload_scene(golden_scenes["book1"])
        c.set_camera(ra.book_camera(cols, rows))
        ctxs.append(c)
        bufs.append(DeviceBuffer(hip, cols * rows * 3))
    ctxs[0].render_noalloc(cols, rows, 0, 2, 50, 1, bufs[0].ptr.value)          # warm both up, one after the other
    ctxs[1].render_noalloc(cols, rows, 0, 2, 50, 1, bufs[1].ptr.value)
    gate, errors = threading.Barrier(2), [None, None]

    def work(k):
        gate.wait()
        try:
            ctxs[k].render_noalloc(cols, rows, 0, spp, 50, 1, bufs[k].ptr.value)
        except ra.RtxError as e:
            errors[k] = str(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    refused = [e for e in errors if e is not None]
    assert len(refused) == 1 and "another rtx_render_noalloc" in refused[0] and "-3" in refused[0], errors
    winner = errors.index(None)
    ctxs[winner].render(cols, rows, 0, spp, 50, 1, want_stats=False)
    np.testing.assert_allclose(bufs[winner].read(), ctxs[winner].read_accum(), rtol=1e-5, atol=1e-5)
    ctxs[1 - winner].render_noalloc(cols, rows, 0, 3, 50, 1, bufs[1 - winner].ptr.value)      # the refused context works again afterwards
    for c, b in zip(ctxs, bufs):
        b.free()
        c.close()
