import sys, os
sys.path.insert(0, os.getcwd())
import raytracing_amd as ra
ctx = ra.Context(0); ctx.upload_scene(ra.random_scene(1)); ctx.set_camera(ra.book_camera(1200, 800))
for flags, name in ((0, "iterative"), (ra.FLAG_RECURSIVE, "recursive")):
    ctx.render(1200, 800, 0, 8, 50, 1, flags=flags)
    best = 1e9
    for _ in range(3):
        _, st = ctx.render(1200, 800, 0, 200, 50, 1, flags=flags)
        best = min(best, ctx.last_trace_ms())
    print(name, "trace_ms %.2f" % best, "rays/sample %.3f" % (st["rays"] / st["samples"]), "Mrays/s %.0f" % (st["rays"] / best / 1e3))
