"""Where does a bench step's time go besides the trace kernel?  1200x800 book-1.  python tools/step_breakdown.py [spp]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
cols, rows = 1200, 800
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 62
ctx = ra.Context(0); ctx.upload_scene(ra.random_scene(1)); ctx.set_camera(ra.book_camera(cols, rows))
for _ in range(3):
    ctx.render(cols, rows, 0, spp, 50, 1, want_stats=False); ctx.resolve(spp)
N = 20
t_enq = t_sync = t_res = 0.0
trace = []
t00 = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter(); ctx.render_async(cols, rows, 0, spp, 50, 1); t1 = time.perf_counter()
    ctx.sync(); t2 = time.perf_counter()
    fb = ctx.resolve(spp); t3 = time.perf_counter()
    t_enq += t1 - t0; t_sync += t2 - t1; t_res += t3 - t2
    trace.append(ctx.last_trace_ms())
tot = (time.perf_counter() - t00) / N * 1e3
print("spp %d: step %.3f ms = enqueue %.3f + wait for trace/reduce %.3f + resolve (kernel, wait, 3.84 MB to host, numpy alloc) %.3f; trace kernel %.3f ms" % (spp, tot, t_enq / N * 1e3, t_sync / N * 1e3, t_res / N * 1e3, sum(trace) / N))
# pinned framebuffer: the resolve kernel writes the caller's memory itself
fbp = ctx.host_framebuffer(cols * rows)
ref = ctx.resolve(spp)
ctx.resolve(spp, out=fbp)
assert (fbp == ref).all()
t00 = time.perf_counter()
for _ in range(N):
    ctx.render_async(cols, rows, 0, spp, 50, 1); ctx.resolve(spp, out=fbp)
print("  render_async + resolve into a pinned framebuffer: %.3f ms per step" % ((time.perf_counter() - t00) / N * 1e3))
t0 = time.perf_counter()
for _ in range(N):
    ctx.resolve(spp, out=fbp)
print("  resolve alone, pinned: %.3f ms; " % ((time.perf_counter() - t0) / N * 1e3), end="")
fbu = np.zeros(cols * rows, dtype=np.uint32)
t0 = time.perf_counter()
for _ in range(N):
    ctx.resolve(spp, out=fbu)
print("pageable: %.3f ms" % ((time.perf_counter() - t0) / N * 1e3))
# the same without the sync in between (what bench.py's step does)
t00 = time.perf_counter()
for _ in range(N):
    ctx.render_async(cols, rows, 0, spp, 50, 1); fb = ctx.resolve(spp)
print("  render_async + resolve back to back: %.3f ms per step" % ((time.perf_counter() - t00) / N * 1e3))
ctx.close()
