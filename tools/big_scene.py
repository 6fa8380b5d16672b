"""A 100 000-sphere scene (RayTracing.cpp:182-213 with grid_half 158) with the uniform grid; the linear list for comparison at 4 spp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
cols, rows = 1200, 800
table = ra.random_scene(1, 158)
for kind, name, spp in ((ra.FILTER_GRID, "grid", 100), (ra.FILTER_MFMA, "linear list", 4)):
    ctx = ra.Context(0)
    ctx.set_filter(kind)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    ctx.render(cols, rows, 0, 1, 50, 1)
    _, st = ctx.render(cols, rows, 0, spp, 50, 1, flags=0)
    ms = ctx.last_trace_ms()
    print("n = %d  %-12s %3d spp  trace %8.2f ms  %8.1f Msamples/s  exact tests/ray %.2f" % (len(table), name, spp, ms, st["samples"] / ms / 1e3, st["candidates"] / st["rays"]))
    ctx.close()
