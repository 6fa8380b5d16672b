"""VERDICT r2 #6(a), a two-level linear list: one bounding sphere per group of g Morton-adjacent spheres in front of the MFMA
filter (1/g of the filter rows), the members of every flagged group straight to the exact test.  How many exact tests per ray
would that cost?  CPU estimate with numpy on the book-1 table (no GPU): primary rays of the 1200x800 camera plus bounce rays
(origins on sphere surfaces, cosine-ish directions), the plain discriminant test against spheres and against group bounds.

    python tools/two_level_estimate.py            (needs librtx.so for the scene generator; no GPU call is made)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracing_amd as ra

rng = np.random.default_rng(7)
t = ra.random_scene(1).astype(np.float64)
C, R = t[:, 0:3], t[:, 3]
n = len(t)


def morton_order(c):
    q = np.clip(((c - c.min(0)) / (np.ptp(c, 0) + 1e-9) * 1023).astype(np.int64), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")


def hits(o, d, c, r):
    """rays x spheres: true where the line's discriminant is positive and a root lies ahead (what the filter flags, roughly)"""
    oc = o[:, None, :] - c[None, :, :]
    a = (d * d).sum(1)[:, None]
    b = (oc * d[:, None, :]).sum(2)
    cc = (oc * oc).sum(2) - (r * r)[None, :]
    disc = b * b - a * cc
    return (disc > 0) & ((-b + np.sqrt(np.maximum(disc, 0))) > 0)


# rays: primaries + bounces
cam = ra.book_camera(1200, 800).c
org = np.array(cam.origin[:]); lc = np.array(cam.left_corner[:]); hor = np.array(cam.horizontal[:]); ver = np.array(cam.vertical[:])
N = 6000
u, v = rng.random(N), rng.random(N)
o1 = np.repeat(org[None], N, 0); d1 = lc[None] + u[:, None] * hor[None] + v[:, None] * ver[None] - org[None]
idx = rng.integers(0, n, N); nrm = rng.normal(size=(N, 3)); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
nrm[idx == 0] = np.abs(nrm[idx == 0]) * np.array([1, 1, 1]); nrm[idx == 0, 1] = np.abs(nrm[idx == 0, 1])
o2 = C[idx] + nrm * R[idx, None] * 1.0001
o2[idx == 0] = np.stack([rng.uniform(-11, 11, (idx == 0).sum()), np.zeros((idx == 0).sum()) + 1e-4, rng.uniform(-11, 11, (idx == 0).sum())], 1)
rd = rng.normal(size=(N, 3)); rd /= np.linalg.norm(rd, axis=1)[:, None]
d2 = nrm + rd * 0.999
d2[idx == 0] = np.array([0, 1, 0]) + rd[idx == 0] * 0.999
o = np.concatenate([o1, o2]); d = np.concatenate([d1, d2])
base = hits(o, d, C, R)
print("spheres %d, rays %d (half primary, half bounce): flagged spheres per ray %.2f (the kernel's exact tests per ray: 1.95)" % (n, len(o), base.sum(1).mean()))
small = np.arange(1, n - 3)                       # the ground and the three unit spheres stay single rows
order = small[morton_order(C[small])]
for g in (2, 4):
    groups = [order[i:i + g] for i in range(0, len(order), g)]
    gc = np.array([C[m].mean(0) for m in groups]); gr = np.array([max(np.linalg.norm(C[k] - C[m].mean(0)) + R[k] for k in m) for m in groups])
    gh = hits(o, d, gc, gr)
    sizes = np.array([len(m) for m in groups])
    tests = (gh * sizes[None, :]).sum(1) + base[:, [0, n - 3, n - 2, n - 1]].sum(1)
    rows = len(groups) + 4
    print("groups of %d: %3d filter rows (%.0f %% of %d), flagged groups per ray %.2f -> exact tests per ray %.2f (x%.1f)"
          % (g, rows, 100.0 * rows / n, n, gh.sum(1).mean(), tests.mean(), tests.mean() / base.sum(1).mean()))
