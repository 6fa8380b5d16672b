"""BASELINE configuration 3 at full size on ONE GPU: 3840x2160, 4096 spp, depth 50 -- a size check (counters, item indices,
partial sums) and a linearity property: two launches of 2048 spp accumulate to the one of 4096 within float summation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra
cols, rows, spp = 3840, 2160, 4096
ctx = ra.Context(0); ctx.upload_scene(ra.random_scene(1)); ctx.set_camera(ra.book_camera(cols, rows))
t0 = time.time(); _, st = ctx.render(cols, rows, 0, spp, 50, 1); wall = time.time() - t0
print("one launch: %.2f s, %d samples (expected %d), %.3f rays/sample, %.0f Msamples/s, term sum ok %s" % (
    wall, st["samples"], cols * rows * spp, st["rays"] / st["samples"], st["samples"] / ctx.last_trace_ms() / 1e3,
    st["term_miss"] + st["term_absorb"] + st["term_depth"] == st["samples"]), flush=True)
full = ctx.read_accum()
ctx.render(cols, rows, 0, spp // 2, 50, 1, want_stats=False)
ctx.render(cols, rows, spp // 2, spp // 2, 50, 1, flags=ra.FLAG_ACCUMULATE, want_stats=False)
two = ctx.read_accum()
rel = np.abs(two - full).max() / np.abs(full).max()
fb1, fb2 = ctx.resolve(spp), None
ctx.write_accum(cols, rows, full); fb2 = ctx.resolve(spp)
print("two halves vs one launch: max rel diff %.2e; 8-bit images differ in %d of %d channels" % (rel, int(((fb1 ^ fb2) != 0).sum()), fb1.size))
assert st["samples"] == cols * rows * spp and rel < 5e-5     # two chunkings of 4096 samples: float summation order
ctx.close()
