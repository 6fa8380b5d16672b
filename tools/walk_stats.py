"""Diagnostic (needs a -DRTX_WALK_STATS build: tools/build_variant.sh wt walkstats -DRTX_WALK_STATS, RTX_LIB_PATH=build/exp/librtx_walkstats.so,
RTX_PRINT_WALK=1): how the grid walk's loop trips are spent -- rays walking per trip, how often each of the trip's three blocks (cell load, exact
test, step) runs and for how many lanes, and the distributions of trips / cells / tests per ray.   python tools/walk_stats.py [spp]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cols, rows = 1200, 800
for name, table in (("book1", ra.random_scene(1)), ("glass", ra.random_scene(1, 11, 0.1, 0.15)), ("20164 spheres", ra.random_scene(1, 71))):
    ctx = ra.Context(0)
    ctx.set_filter(ra.FILTER_GRID)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    print("== %s, %d spp" % (name, spp), file=sys.stderr, flush=True)
    _, st = ctx.render(cols, rows, 0, spp, 50, 1)
    print("   trace %.2f ms, %.2f rays per sample, %.2f tests per ray" % (ctx.last_trace_ms(), st["rays"] / st["samples"], st["candidates"] / st["rays"]), file=sys.stderr, flush=True)
    ctx.close()
