"""Cost-ordered work lists against the geometric chunk plan (trace-kernel ms from HIP events, 1200x800 book-1, one MI355X).

    python tools/order_ab.py [spp ...]            env: RTX_FILTER, RTX_ORDER_CHUNKS, RTX_ORDER_SHARE_PCT

For every spp: `plain` = ordering off (RTX_ORDER=0 semantics through rtx_set_tuning), `ordered` = the third launch of the same
frame (the first measures the pixels' costs, the second sorts and runs ordered and measures again).  Also checks that the
accumulation buffers are bit-identical when both launches use the same chunk boundaries."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra

cols, rows = 1200, 800
spps = [int(a) for a in sys.argv[1:]] or [62, 500]
scene = os.environ.get("SCENE", "book1")
table = ra.random_scene(1) if scene == "book1" else (ra.random_scene(1, 11, 0.1, 0.15) if scene == "glass" else ra.random_scene(1, 71))
for spp in spps:
    ctx = ra.Context(0)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    ctx.set_tuning("order", 0)
    ctx.render(cols, rows, 0, 4, 50, 1)
    plain = 1e9
    for _ in range(3):
        _, st = ctx.render(cols, rows, 0, spp, 50, 1)
        plain = min(plain, ctx.last_trace_ms())
    plan_plain = ctx.last_chunk_plan()[1]
    ctx.set_tuning("order", 1)
    ordered = 1e9
    times = []
    for _ in range(5):
        _, st = ctx.render(cols, rows, 0, spp, 50, 1)
        times.append(ctx.last_trace_ms())
    ordered = min(times[1:])
    plan_ord = ctx.last_chunk_plan()[1]
    a_ord = ctx.read_accum()
    # same boundaries, unordered: must be the same bits
    ctx.set_tuning("order", 0)
    ctx2 = ra.Context(0)
    ctx2.upload_scene(table)
    ctx2.set_camera(ra.book_camera(cols, rows))
    ctx2.set_tuning("order", 0)
    same = None
    sizes = np.diff(plan_ord)
    if len(set(sizes.tolist())) == 1:
        ctx2.render(cols, rows, 0, spp, 50, 1, chunk_spp=int(sizes[0]))
        same = bool((ctx2.read_accum().view(np.uint32) == a_ord.view(np.uint32)).all())
    ctx2.close()
    print("%s spp %4d  plain %.3f ms (%d chunks)  ordered %.3f ms (%d chunks, first launch %.3f)  %.0f -> %.0f Msamples/s  same bits as uniform chunks: %s"
          % (scene, spp, plain, len(plan_plain) - 1, ordered, len(plan_ord) - 1, times[0], st["samples"] / plain / 1e3, st["samples"] / ordered / 1e3, same), flush=True)
    ctx.close()
