import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
cols, rows, spp = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 100
table = ra.random_scene(1)
ctx = ra.Context(0)
ctx.upload_scene(table)
ctx.set_camera(ra.book_camera(cols, rows))
ctx.render(cols, rows, 0, 8, 50, 1)
best = 1e9
for rep in range(3):
    _, st = ctx.render(cols, rows, 0, spp, 50, 1)
    best = min(best, ctx.last_trace_ms())
print(os.environ.get("RTX_LIB_PATH", "default"), "block", os.environ.get("RTX_BLOCK", "-"), "filter", os.environ.get("RTX_FILTER", "mfma"),
      "trace_ms %.2f" % best, "cand/ray %.3f" % (st["candidates"] / st["rays"]), "Msamples/s %.0f" % (st["samples"] / best / 1e3))
ctx.close()
