#!/usr/bin/env python3
"""bench_ref_vs_port.py -- TEST INFRASTRUCTURE, container only (needs oracle/_ref, i.e. /root/reference).

SURVEY s8(d) asks that the CPU baseline bench.py times on the GPU box -- the oracle's tile-pool renderer ro_render, a port of
the reference CPU path -- be validated against the compiled reference "within a stated band at equal thread count".  This
script states the band: it times the reference's UNMODIFIED renderScene (raytracer.cpp:60-143 through oracle/_ref/ref_harness
render_ref, an external steady_clock around the call; mind its 1 s completion poll at raytracer.cpp:130-133, which rounds
every reference time up by as much as a second -- runs are sized to >= 20 s) against ro_render on BASELINE configuration 1
(book-1, 1200x800, depth 50, 64x64 tile jobs) at 1 thread and at all of this container's cores, compares the two images
(PSNR of the 8-bit frames at equal spp; both are Monte Carlo renders from different random streams) and writes
oracle/ref_vs_port.json, which bench.py quotes as cpu_baseline.port_vs_reference (a constant with this provenance).

    python oracle/bench_ref_vs_port.py [--spp1 2] [--sppN 10]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pyoracle  # noqa: E402

COLS, ROWS, DEPTH, TILE = 1200, 800, 50, 64


def time_reference(scene_bin, spp, threads, out_dir):
    fb_path = os.path.join(out_dir, "ref_%d.u32" % threads)
    cmd = [os.path.join(HERE, "_ref", "ref_harness"), "render_ref", scene_bin, fb_path, str(COLS), str(ROWS), str(spp), str(DEPTH), str(threads), str(TILE), "0", "12345"]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, check=True)
    wall = time.perf_counter() - t0
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    return d["seconds"], wall, np.fromfile(fb_path, dtype=np.uint32)


def time_port(table, spp, threads):
    cam = pyoracle.book_camera(COLS, ROWS)
    t0 = time.perf_counter()
    accum, _, st = pyoracle.render(table, cam, ROWS, COLS, 0, spp, DEPTH, 1, nthreads=threads, tile=TILE)
    dt = time.perf_counter() - t0
    return dt, pyoracle.resolve(accum, ROWS, COLS, spp), st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp1", type=int, default=2, help="samples per pixel of the 1-thread runs")
    ap.add_argument("--sppN", type=int, default=10, help="samples per pixel of the all-cores runs (configuration 1: 10)")
    args = ap.parse_args()
    if not os.path.exists(os.path.join(HERE, "_ref", "ref_harness")):
        raise SystemExit("oracle/_ref/ref_harness is missing: `make -C oracle ref` (needs /root/reference)")
    cores = os.cpu_count() or 1
    table = pyoracle.random_scene(1)
    out = {"config": "book-1 (seed 1, %d spheres) %dx%d, depth %d, %dx%d tile jobs" % (len(table), COLS, ROWS, DEPTH, TILE, TILE),
           "host": {"cores": cores, "cpu_model": next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?")},
           "reference": "unmodified renderScene (raytracer.cpp:60-143), g++ -O2 -ffp-contract=off, steady_clock around the call (includes its <= 1 s completion poll)",
           "port": "oracle/rt_oracle.c ro_render, gcc -O2 -ffp-contract=off", "runs": []}
    with tempfile.TemporaryDirectory() as tmp:
        scene_bin = os.path.join(tmp, "scene.bin")
        pyoracle.write_scene_bin(scene_bin, table)
        for threads, spp in ((1, args.spp1), (cores, args.sppN)):
            ref_s, ref_wall, ref_fb = time_reference(scene_bin, spp, threads, tmp)
            port_s, port_fb, st = time_port(table, spp, threads)
            samples = COLS * ROWS * spp
            run = {"threads": threads, "spp": spp,
                   "reference_seconds": round(ref_s, 3), "reference_msamples_per_s": round(samples / ref_s / 1e6, 4),
                   "reference_msamples_per_s_poll_corrected_upper": round(samples / max(ref_s - 1.0, 1e-3) / 1e6, 4),
                   "port_seconds": round(port_s, 3), "port_msamples_per_s": round(samples / port_s / 1e6, 4),
                   "port_over_reference": round(ref_s / port_s, 3),
                   "psnr_db_8bit_frames": round(float(pyoracle.psnr_rgb8(ref_fb, port_fb)[0]), 2),
                   "port_rays_per_sample": round(st["rays"] / st["samples"], 4)}
            out["runs"].append(run)
            print(json.dumps(run), flush=True)
    with open(os.path.join(HERE, "ref_vs_port.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
