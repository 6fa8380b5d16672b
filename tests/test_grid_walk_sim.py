"""The uniform grid of RTX_FILTER_GRID on the CPU: a float32 restatement of the kernel's walk (trace_kernel.hip, kFilterGrid) over the
grid that the library builds (rtx_build_grid), against the reference's linear scan (raytracer.cpp:285-304 / sphere.cpp:68-102) ray by
ray.  It checks the construction and the walk's logic -- tiers, big list, early stop, far origins -- where no GPU is needed; the
GPU tests check the kernel itself."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ra():
    import raytracing_amd
    raytracing_amd.lib()
    return raytracing_amd

f32 = np.float32
FLT_MAX = f32(3.402823466e+38)


def _dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))       # vector_cuda.h:68-71


def _sphere_t(o, d, a, c, r):
    """sphere.cpp:72-99: the accepted root, or None."""
    oc = (o - c).astype(f32)
    bq = _dot(oc, d)
    cq = f32(_dot(oc, oc) - f32(r * r))
    disc = f32(f32(bq * bq) - f32(a * cq))
    if not disc > 0:
        return None
    sq = f32(np.sqrt(disc))
    t = f32(f32(-bq - sq) / a)
    if not t > f32(0.001):
        t = f32(f32(-bq + sq) / a)
    return t if (t > f32(0.001) and t < FLT_MAX) else None


def _brute(table, o, d):
    a = _dot(d, d)
    best = (FLT_MAX, 0xFFFFFFFF)
    for i in range(len(table)):
        t = _sphere_t(o, d, a, table[i, 0:3], table[i, 3])
        if t is not None and (t, i) < best:
            best = (t, i)
    return best


def _walk(table, g, o, d):
    """The kernel's phase (C) for one ray; returns ((t, index), how the ray was served)."""
    a = _dot(d, d)
    best = [FLT_MAX, 0xFFFFFFFF]

    def test(i):
        t = _sphere_t(o, d, a, table[i, 0:3], table[i, 3])
        if t is not None and (t < best[0] or (t == best[0] and i < best[1])):
            best[0], best[1] = t, i

    for i in g["big"]:
        test(int(i))
    dim = g["dim"]
    if dim[0] == 0:
        return tuple(best), "no grid"
    gmin, h, inv_h, pad = g["gmin"], g["h"], g["inv_h"], f32(g["pad"])
    slack_t = f32(f32(2) * pad / f32(np.sqrt(a)))
    lo = gmin
    hi = (gmin + dim.astype(f32) * h).astype(f32)
    off = np.maximum(np.maximum(lo - o, o - hi), f32(0)).astype(f32)
    l_far = f32(f32(np.sqrt(f32(off @ off))) * f32(1.00001) + g["box_diag"])
    far = not (l_far <= g["tier_s"][2])
    rm = g["r_min"]
    extra = f32(pad + (f32(np.sqrt(f32(rm * rm + f32(2.0 ** -19) * f32(l_far * l_far + rm * rm)))) * f32(1.00001) - rm)) if far else f32(0)
    t_in, t_out, walking = f32(0), f32(np.inf), True
    inv = np.zeros(3, f32)
    for k in range(3):
        if d[k] != 0:
            inv[k] = f32(1) / d[k]
            ta, tb = f32((lo[k] - extra - o[k]) * inv[k]), f32((hi[k] + extra - o[k]) * inv[k])
            t_in, t_out = max(t_in, min(ta, tb)), min(t_out, max(ta, tb))
        elif o[k] < lo[k] - extra or o[k] > hi[k] + extra:
            walking = False
    walking = walking and t_in <= t_out
    if walking and far:
        for i in range(len(table)):
            test(i)
        return tuple(best), "scan"
    if not walking:
        return tuple(best), "missed the box"
    tn, dt, cs, rem, cidx = np.zeros(3, f32), np.zeros(3, f32), [0, 0, 0], [0, 0, 0], 0
    stride = [1, int(dim[0]), int(dim[0] * dim[1])]
    for k in range(3):
        p = f32(o[k] + t_in * d[k])
        c = int(np.floor(f32((p - gmin[k]) * inv_h[k])))
        c = min(max(c, 0), int(dim[k]) - 1)
        cidx += c * stride[k]
        fwd = d[k] > 0
        if d[k] != 0:
            tn[k] = f32(f32(f32(gmin[k] + f32(f32(c + (1 if fwd else 0)) * h[k])) - o[k]) * inv[k])
            dt[k] = f32(h[k] * abs(inv[k]))
            rem[k] = (int(dim[k]) - 1 - c) if fwd else c
        else:
            tn[k] = f32(np.inf)
        cs[k] = stride[k] if fwd else -stride[k]
    s_max = f32(min(t_out, f32(best[0] + slack_t)) * f32(f32(np.sqrt(a)) * f32(1.00001)))
    tier = 2 if not (s_max <= g["tier_s"][1]) else (1 if not (s_max <= g["tier_s"][0]) else 0)
    start, tend, items = g["cell_start"], g["tier_end"], g["items"]
    while True:
        for it in range(int(start[cidx]), int(tend[cidx, tier])):
            test(int(items[it]))
        t_leave = min(tn)
        ax = 0 if (tn[0] <= tn[1] and tn[0] <= tn[2]) else (1 if tn[1] <= tn[2] else 2)
        if f32(f32(best[0] + slack_t) + f32(best[0] * f32(4.0e-6))) < t_leave or rem[ax] == 0:
            break
        rem[ax] -= 1
        cidx += cs[ax]
        tn[ax] = f32(tn[ax] + dt[ax])
    return tuple(best), "tier %d" % tier


def _grid(ra, table):
    g = ra.build_grid(table)
    g["dim"] = g["dim"].astype(np.int64)
    return g


def _scene(rng, n, spread, rmax, tiny=False):
    t = np.zeros((n, 10), dtype=f32)
    t[:, 0] = rng.uniform(-spread, spread, n)
    t[:, 1] = rng.uniform(0.0, 0.3, n)
    t[:, 2] = rng.uniform(-spread, spread, n)
    t[:, 3] = rng.uniform(0.05, rmax, n)
    if tiny:
        t[::5, 3] = 1.0e-3
        t[3::11, 3] = 0.0
    t[0, 0:4] = (0, -1000, 0, 1000)              # the ground
    t[1, 0:4] = (0, 1.5, 0, 1.5)                 # one radius outlier
    t[:, 4] = 1
    return t


@pytest.mark.parametrize("case", ["field", "tiny", "wide_field"])
def test_walk_equals_linear_scan(ra, case):
    rng = np.random.default_rng({"field": 1, "tiny": 2, "wide_field": 3}[case])
    table = _scene(rng, 260 if case != "wide_field" else 420, 8.0 if case != "wide_field" else 60.0, 0.3, tiny=(case == "tiny"))
    g = _grid(ra, table)
    served = {}
    for k in range(260):
        kind = k % 4
        if kind == 0:                                # from a camera beside the field
            o = np.array([rng.uniform(9, 14), rng.uniform(0.5, 3), rng.uniform(-4, 4)], dtype=f32)
            d = (np.array([rng.uniform(-6, 6), rng.uniform(0, 0.4), rng.uniform(-6, 6)], dtype=f32) - o).astype(f32)
        elif kind == 1:                              # a bounce from a sphere's surface
            i = int(rng.integers(2, len(table)))
            nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
            o = (table[i, 0:3] + table[i, 3] * nrm.astype(f32)).astype(f32)
            d = (nrm + 0.9 * rng.normal(size=3)).astype(f32)
        elif kind == 2:                              # skimming along the field
            o = np.array([rng.uniform(-7, 7), rng.uniform(0.05, 0.5), rng.uniform(-7, 7)], dtype=f32)
            d = np.array([rng.normal(), rng.normal() * 0.02, rng.normal()], dtype=f32)
        else:                                        # from far away, towards the field (some beyond the registrations' reach)
            far = 10.0 ** rng.uniform(1.5, 4.5)
            o = (np.array([rng.normal(), abs(rng.normal()) * 0.3 + 0.05, rng.normal()]) * far).astype(f32)
            aim = np.array([rng.uniform(-6, 6), rng.uniform(0, 0.4), rng.uniform(-6, 6)], dtype=f32)
            d = ((aim - o) / f32(far)).astype(f32)
        with np.errstate(over="ignore", invalid="ignore"):      # FLT_MAX + slack is inf, as on the GPU
            got, how = _walk(table, g, o, d)
        served[how] = served.get(how, 0) + 1
        assert got == _brute(table, o, d), (case, k, how, o, d)
    assert served.get("tier 0", 0) > 50 and served.get("scan", 0) > 5, served
