"""CPU model of lane decoupling in the grid walk, fed with the measured distribution of loop trips per ray
(tools/walk_stats.py on a -DRTX_WALK_STATS build, gpurun_out/walk_stats50.txt) and the measured share of the walk in the kernel's time
(-DRTX_PHASE_TIMERS).  One wave = 64 lanes; an iteration costs F (everything but the walk loop, in units of one loop trip) + the trips it
runs; a lane whose ray has finished its walk is shaded and gets its next ray in the next iteration's fixed part.  Policies: the kernel's
(walk until every ray is done), a fixed number of trips per iteration, a threshold on the rays still walking, both.
    python tools/decouple_sim.py gpurun_out/walk_stats50.txt"""
import sys

import numpy as np

F_BY_SCENE = {"book1": (55.7 / 44.3) * 8.7, "glass": (57.0 / 43.0) * 8.4, "20164 spheres": (34.9 / 65.1) * 21.6}


def simulate(hist, F, policy, iters=20000, seed=1):
    rng = np.random.default_rng(seed)
    p = hist / hist.sum()
    vals = np.arange(len(hist))
    vals[-1] = 90                                  # the open last bucket
    rem = rng.choice(vals, size=64, p=p)
    cost = 0.0; rays = 0; trips_total = 0
    for _ in range(iters):
        k = 0
        while True:
            walking = int((rem > 0).sum())
            if policy(k, walking):
                break
            rem[rem > 0] -= 1
            k += 1
        done = rem == 0
        n = int(done.sum())
        rays += n
        rem[done] = rng.choice(vals, size=n, p=p)
        cost += F + k; trips_total += k
    return cost / rays, trips_total / iters, rays / iters


def main():
    scenes = {}
    name = None
    for ln in open(sys.argv[1]):
        if ln.startswith("=="):
            name = ln[3:].split(",")[0].strip()
        elif "rays by trips" in ln:
            scenes[name] = np.array([float(x) for x in ln.split(":")[1].split()])
    for name, hist in scenes.items():
        F = F_BY_SCENE[name]
        base, trips, rays = simulate(hist, F, lambda k, w: w == 0)
        print("== %s: F = %.1f trips; kernel's policy: %.1f trips and %.1f rays per iteration (model), cost per ray %.4f" % (name, F, trips, rays, base))
        best = []
        for K in (2, 3, 4, 5, 6, 8, 10, 12, 16):
            c, t, r = simulate(hist, F, lambda k, w, K=K: w == 0 or k >= K)
            best.append((c / base, "at most %d trips" % K, t, r))
        for T in (1, 2, 4, 8, 12, 16, 24, 32):
            c, t, r = simulate(hist, F, lambda k, w, T=T: w <= T and (w == 0 or k > 0))
            best.append((c / base, "until <= %d rays walk" % T, t, r))
        for K in (2, 4, 6, 8, 12):
            for T in (4, 8, 16, 24):
                c, t, r = simulate(hist, F, lambda k, w, K=K, T=T: w == 0 or (k >= K and w <= T))
                best.append((c / base, ">= %d trips, then until <= %d rays walk" % (K, T), t, r))
        best.sort()
        for rel, what, t, r in best[:8]:
            print("   %-46s cost per ray x %.3f   (%.1f trips, %.1f rays per iteration)" % (what, rel, t, r))


if __name__ == "__main__":
    main()
