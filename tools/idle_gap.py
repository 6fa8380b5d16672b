"""How much does a launch pay for starting on an idle GPU?  62-spp launches of the bench frame with a host-side pause between
the end of one and the enqueue of the next (trace-kernel ms from HIP events).  python tools/idle_gap.py [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
cols, rows = 1200, 800
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 62
ctx = ra.Context(0); ctx.upload_scene(ra.random_scene(1)); ctx.set_camera(ra.book_camera(cols, rows))
for _ in range(3):
    ctx.render(cols, rows, 0, spp, 50, 1, want_stats=False)
for gap_us in (0, 20, 50, 100, 200, 500, 1000, 3000, 10000):
    ms = []
    for _ in range(6):
        ctx.render_async(cols, rows, 0, spp, 50, 1)
        ctx.sync()
        t_end = time.perf_counter()
        while (time.perf_counter() - t_end) * 1e6 < gap_us:
            pass
        ctx.render_async(cols, rows, 0, spp, 50, 1)
        ctx.sync()
        ms.append(ctx.last_trace_ms())
    print("pause %6d us: trace %.3f ms (min of 6; max %.3f)" % (gap_us, min(ms), max(ms)), flush=True)
# no pause at all: queued behind the previous launch
for k in range(4):
    ctx.render_async(cols, rows, 0, spp, 50, 1)
ctx.sync()
print("queued behind the previous launch: %.3f ms" % ctx.last_trace_ms())
ctx.close()
