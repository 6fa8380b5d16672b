"""One render of BASELINE configuration 5 (20 164 spheres, table read through L2) for counter passes: tools/pmc_c5.sh."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
table = ra.random_scene(1, 71)
ctx = ra.Context(0)
ctx.upload_scene(table)
ctx.set_camera(ra.book_camera(1200, 800))
ctx.render(1200, 800, 0, 1, 50, 1)
_, st = ctx.render(1200, 800, 0, spp, 50, 1)
ms = ctx.last_trace_ms()
print("C5 n=%d spp=%d trace %.2f ms %.1f Msamples/s cand/ray %.2f" % (len(table), spp, ms, st["samples"] / ms / 1e3, st["candidates"] / st["rays"]))
ctx.close()
