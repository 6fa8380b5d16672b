"""Runs the trace kernel with RTX_FLAG_VALIDATE_FILTER (every ray also takes the reference's full ascending scan) over
several scenes, cameras and seeds; any filter miss is an error.  python tools/validate_filter.py [spp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cols, rows = 1200, 800
cases = [("book seed 1", ra.random_scene(1), None), ("book seed 7", ra.random_scene(7), None), ("glass heavy", ra.random_scene(3, 11, 0.1, 0.95), None),
         ("metal heavy", ra.random_scene(5, 11, 0.1, 0.9), None), ("1.6k spheres", ra.random_scene(2, 20), None), ("5k spheres", ra.random_scene(2, 35), None)]
bad = 0
for name, table, _ in cases:
    for filt in (ra.FILTER_MFMA, ra.FILTER_VALU):
        ctx = ra.Context(0)
        ctx.set_filter(filt)
        ctx.upload_scene(table)
        ctx.set_camera(ra.book_camera(cols, rows))
        _, st = ctx.render(cols, rows, 0, spp, 50, 11, flags=ra.FLAG_VALIDATE_FILTER)
        print("%-14s filter %d n=%5d rays %11d cand/ray %.3f filter_misses %d" % (name, filt, len(table), st["rays"], st["candidates"] / st["rays"], st["filter_misses"]), flush=True)
        bad += st["filter_misses"]
        ctx.close()
print("TOTAL filter misses:", bad)
sys.exit(1 if bad else 0)
