"""Msamples/s of the MFMA filter (linear list) and the uniform grid over scene sizes: where RTX_FILTER_AUTO should switch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
cols, rows, spp = 1200, 800, 100
for gh in (2, 3, 4, 5, 6, 7, 9, 11, 14, 20):
    table = ra.random_scene(1, gh)
    row = []
    for kind in (ra.FILTER_MFMA, ra.FILTER_GRID):
        ctx = ra.Context(0)
        ctx.set_filter(kind)
        ctx.upload_scene(table)
        ctx.set_camera(ra.book_camera(cols, rows))
        ctx.render(cols, rows, 0, 8, 50, 1)
        best = 1e9
        for _ in range(2):
            _, st = ctx.render(cols, rows, 0, spp, 50, 1)
            best = min(best, ctx.last_trace_ms())
        row.append(st["samples"] / best / 1e3)
        ctx.close()
    print("grid_half %2d  n = %5d   mfma %8.0f   grid %8.0f   Msamples/s" % (gh, len(table), row[0], row[1]))
