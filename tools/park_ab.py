"""Lane decoupling in the grid walk (tuning knob walk_park): trace-kernel ms by threshold, and a bit-for-bit comparison of
the accumulation buffers with the undecoupled walk.    python tools/park_ab.py [spp] [thresholds ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
parks = [int(a) for a in sys.argv[2:]] or [0, 4, 8, 12, 16, 24]
cols, rows = 1200, 800
for name, table in (("book1", ra.random_scene(1)), ("glass", ra.random_scene(1, 11, 0.1, 0.15)), ("20164 spheres", ra.random_scene(1, 71))):
    ref = None
    line = "%-14s %4d spp " % (name, spp)
    for park in parks:
        ctx = ra.Context(0)
        ctx.set_filter(ra.FILTER_GRID)
        try:
            ctx.set_tuning("walk_park", park)
        except Exception:
            if park:
                continue
        ctx.upload_scene(table)
        ctx.set_camera(ra.book_camera(cols, rows))
        ctx.set_tuning("order", 0)
        ctx.render(cols, rows, 0, 4, 50, 1)
        plain = 1e9
        for _ in range(3):
            _, st = ctx.render(cols, rows, 0, spp, 50, 1)
            plain = min(plain, ctx.last_trace_ms())
        acc = ctx.read_accum().view(np.uint32)
        if ref is None:
            ref, ref_st = acc, st
        same = bool((acc == ref).all()) and all(st[k] == ref_st[k] for k in ("samples", "rays", "term_miss", "term_absorb", "term_depth", "candidates"))
        ctx.set_tuning("order", 1)
        ordered = 1e9
        for _ in range(4):
            ctx.render(cols, rows, 0, spp, 50, 1)
            ordered = min(ordered, ctx.last_trace_ms())
        line += " | park %2d: %6.2f / %6.2f ms %s" % (park, plain, ordered, "same bits" if same else "DIFFERENT")
        ctx.close()
    print(line, flush=True)
