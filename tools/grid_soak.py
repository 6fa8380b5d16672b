"""usage: grid_soak.py [seconds [seed [grid|mfma|valu]]]
Soak of the uniform grid (or another candidate search) against the full-scan validation (RTX_FLAG_VALIDATE_FILTER): random tables (clouds, fields, mixed radii over
four decades, outliers, duplicates), random cameras (inside, beside, skimming, far away), random densities.  Any miss is printed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracing_amd as ra

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mode = sys.argv[3] if len(sys.argv) > 3 else "grid"      # "pair": the production kernels of grid and mfma against those of valu, frame by frame
filt = {"grid": ra.FILTER_GRID, "mfma": ra.FILTER_MFMA, "valu": ra.FILTER_VALU, "pair": ra.FILTER_GRID}[mode]
t0, runs, rays, misses = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    kind = rng.integers(0, 4)
    n = int(10 ** rng.uniform(0.5, 4.2))
    scale = 10.0 ** rng.uniform(-2, 4)
    t = np.zeros((n, 10), dtype=np.float32)
    if kind == 0:      # cloud
        t[:, 0:3] = rng.normal(size=(n, 3)) * scale
    elif kind == 1:    # field on a plane
        t[:, 0] = rng.uniform(-1, 1, n) * scale; t[:, 2] = rng.uniform(-1, 1, n) * scale; t[:, 1] = rng.uniform(0, 0.01, n) * scale
    elif kind == 2:    # line
        t[:, 0] = rng.uniform(-1, 1, n) * scale; t[:, 1:3] = rng.normal(size=(n, 2)) * scale * 1e-3
    else:              # clusters
        c = rng.normal(size=(8, 3)) * scale
        t[:, 0:3] = c[rng.integers(0, 8, n)] + rng.normal(size=(n, 3)) * scale * 0.03
    t[:, 3] = scale * 10.0 ** rng.uniform(-4, -0.5, n) * rng.choice([1, 1, 1, 1, -1], n)
    if n > 3:
        t[0, 0:4] = (0, -1000 * scale, 0, 1000 * scale)              # a ground
        t[1, 3] = scale * 0.5                                        # an outlier
        t[3] = t[2]                                                  # a duplicate
        if rng.random() < 0.3:
            t[2, 3] = 0.0
    t[:, 4] = rng.integers(1, 4, n); t[:, 5:8] = 0.7; t[:, 8] = rng.uniform(0, 0.3, n); t[:, 9] = 1.5
    ck = rng.integers(0, 4)
    look = t[rng.integers(0, n), 0:3].astype(np.float64)
    if ck == 0:   pos = look + rng.normal(size=3) * scale * 0.2
    elif ck == 1: pos = look + rng.normal(size=3) * scale * 3
    elif ck == 2: pos = look + np.array([rng.normal(), 0.002 * rng.normal(), rng.normal()]) * scale
    else:         pos = look + rng.normal(size=3) * scale * 10.0 ** rng.uniform(1, 4)
    cols, rows = 96, 64
    fov = float(np.clip(np.degrees(2 * np.arctan(scale / max(np.linalg.norm(pos - look), 1e-30))), 0.05, 90.0)) if ck == 3 else float(rng.uniform(10, 90))
    ctx = ra.Context(0)
    try:
        ctx.set_filter(filt)
        ctx.set_tuning("grid_density_pct", int(10 ** rng.uniform(0.7, 2.6)))
        ctx.upload_scene(t)
        ctx.set_camera(ra.Camera(fov, cols / rows, 0.0, 1.0, pos.astype(np.float32), (0, 1, 0), look.astype(np.float32)))
        seed = int(rng.integers(1, 1 << 30))
        if mode == "pair":
            frames = []
            for f in (ra.FILTER_VALU, ra.FILTER_GRID, ra.FILTER_MFMA):
                ctx.set_filter(f)
                _, st = ctx.render(cols, rows, 0, 2, 8, seed, chunk_spp=2)
                frames.append(ctx.read_accum().view(np.uint32))
            nan = np.isnan(frames[0].view(np.float32))
            st["filter_misses"] = int(((frames[1] != frames[0]) & ~nan).sum() + ((frames[2] != frames[0]) & ~nan).sum())
        else:
            _, st = ctx.render(cols, rows, 0, 2, 8, seed, flags=ra.FLAG_VALIDATE_FILTER, chunk_spp=2)
    finally:
        ctx.close()
    runs += 1; rays += st["rays"]; misses += st["filter_misses"]
    if st["filter_misses"]:
        print("MISS", dict(kind=int(kind), n=n, scale=scale, cam=int(ck), misses=st["filter_misses"], rays=st["rays"]), flush=True)
    if runs % 50 == 0:
        print("%d scenes, %d rays, %d misses, %.0f s" % (runs, rays, misses, time.time() - t0), flush=True)
print("soak done: %d scenes, %d rays, %d misses" % (runs, rays, misses))
sys.exit(1 if misses else 0)
