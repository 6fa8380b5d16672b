import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
cols, rows, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 100
table = ra.random_scene(1)
for name, kind in (("valu", ra.FILTER_VALU), ("mfma", ra.FILTER_MFMA)):
    ctx = ra.Context(0)
    ctx.set_filter(kind)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    ctx.render(cols, rows, 0, 8, 50, 1)
    for rep in range(2):
        _, st = ctx.render(cols, rows, 0, spp, 50, 1)
    print(name, "trace_ms %.2f" % ctx.last_trace_ms(), "cand/ray %.3f" % (st["candidates"] / st["rays"]), "rays", st["rays"],
          "Msamples/s %.0f" % (st["samples"] / ctx.last_trace_ms() / 1e3))
    ctx.close()
