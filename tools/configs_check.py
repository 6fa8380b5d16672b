"""Times the BASELINE configurations 2-5 (scaled down in spp where noted) on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra

def run(name, table, cols, rows, spp, depth=50):
    ctx = ra.Context(0)
    ctx.upload_scene(table)
    ctx.set_camera(ra.book_camera(cols, rows))
    ctx.render(cols, rows, 0, 1, depth, 1)
    t0 = time.time()
    _, st = ctx.render(cols, rows, 0, spp, depth, 1)
    wall = time.time() - t0
    ms = ctx.last_trace_ms()
    print("%-34s n=%5d %4dx%-4d spp=%4d  trace %9.2f ms  wall %7.2f s  %8.1f Msamples/s %8.1f Mrays/s  rays/sample %.2f  cand/ray %.2f  frac %.3f" % (
        name, len(table), cols, rows, spp, ms, wall, st["samples"] / ms / 1e3, st["rays"] / ms / 1e3, st["rays"] / st["samples"],
        st["candidates"] / st["rays"], 18.0 * st["sphere_tests"] / (ms * 1e-3) / 157.3e12), flush=True)
    ctx.close()

book = ra.random_scene(1)
run("C2 book1 500spp", book, 1200, 800, 500)
run("C3 book1 4K (64 of 4096 spp)", book, 3840, 2160, 64)
run("C4 glass-heavy 100spp", ra.random_scene(1, 11, 0.1, 0.95), 1200, 800, 100)
run("C5 scaled 20k spheres 100spp", ra.random_scene(1, 71), 1200, 800, 100)
run("mid 1.6k spheres 100spp", ra.random_scene(1, 20), 1200, 800, 100)
run("mid 5k spheres 50spp", ra.random_scene(1, 35), 1200, 800, 50)
