"""Diagnostic (needs a -DRTX_WAVE_TIMES build: tools/build_variant.sh wt wavetimes -DRTX_WAVE_TIMES, RTX_LIB_PATH=build/exp/librtx_wavetimes.so):
when the waves of a launch start, see the work counter dry and end.  python tools/wave_times.py <spp> [order 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
cols, rows = 1200, 800
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 62
order = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = ra.Context(0)
ctx.upload_scene(ra.random_scene(1))
ctx.set_camera(ra.book_camera(cols, rows))
ctx.set_tuning("order", order)
for k in range(3):
    if k == 2:
        os.environ["RTX_PRINT_WAVES"] = "1"
    _, st = ctx.render(cols, rows, 0, spp, 50, 1)
print("spp %d order %d trace %.3f ms, chunks %d" % (spp, order, ctx.last_trace_ms(), len(ctx.last_chunk_plan()[1]) - 1))
ctx.close()
# back-to-back launches (no host wait in between): does an isolated launch pay for a cold start?
ctx = ra.Context(0)
ctx.upload_scene(ra.random_scene(1))
ctx.set_camera(ra.book_camera(cols, rows))
ctx.set_tuning("order", order)
os.environ.pop("RTX_PRINT_WAVES", None)
for k in range(3):
    ctx.render(cols, rows, 0, spp, 50, 1)
iso = ctx.last_trace_ms()
for n in (2, 4, 8):
    for k in range(n):
        ctx.render_async(cols, rows, 0, spp, 50, 1)
    ctx.sync()
    print("  %d launches back to back: last trace %.3f ms, all kernels %.3f ms (isolated %.3f)" % (n, ctx.last_trace_ms(), ctx.last_kernel_ms(), iso))
ctx.close()
