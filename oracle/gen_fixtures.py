#!/usr/bin/env python3
"""Regenerates tests/golden/* from the COMPILED REFERENCE (oracle/_ref/ref_harness).  TEST INFRASTRUCTURE ONLY.

Runs only where /root/reference exists (this container); the fixtures it writes are data -- inputs and the
reference's outputs -- and are committed so that the tests run anywhere.  Usage: python oracle/gen_fixtures.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pyoracle  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "tests", "golden")
HARNESS = os.path.join(HERE, "_ref", "ref_harness")

# name: (scene key, cols, rows, spp_begin, spp_count, max_depth, seed, recursive, keep)
FRAMES = {
    "book1_96x64_spp2":          ("book1", 96, 64, 0, 2, 50, 1, 0, ("primary", "accum", "fb")),
    "book1_96x64_spp2_recursive": ("book1", 96, 64, 0, 2, 50, 1, 1, ("accum", "fb")),
    "book1_96x64_spp2_depth3":   ("book1", 96, 64, 0, 2, 3, 1, 0, ("accum", "fb")),
    "book1_96x64_spp3_from5":    ("book1", 96, 64, 5, 3, 50, 7, 0, ("primary", "accum")),
    "glass_96x64_spp2":          ("glass", 96, 64, 0, 2, 50, 1, 0, ("primary", "accum", "fb")),
    "scaled_64x48_spp1":         ("scaled", 64, 48, 0, 1, 50, 1, 0, ("primary", "accum", "fb")),
    "book1_300x200_spp1":        ("book1", 300, 200, 0, 1, 50, 1, 0, ("primary", "fb")),
    "book1_75x50_spp16":         ("book1", 75, 50, 0, 16, 50, 3, 0, ("accum", "fb")),
}

SCENES = {
    "book1":  dict(seed=1, grid_half=11, p_diffuse=0.8, p_metal=0.95),   # RayTracing.cpp:182-213
    "glass":  dict(seed=1, grid_half=11, p_diffuse=0.1, p_metal=0.95),   # BASELINE config 4: 85 % glass
    "scaled": dict(seed=1, grid_half=20, p_diffuse=0.8, p_metal=0.95),   # a ~1.6k-sphere cut of BASELINE config 5
}


def main():
    if not os.path.exists(HARNESS):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    os.makedirs(GOLDEN, exist_ok=True)
    sizes = json.loads(subprocess.check_output([HARNESS, "sizes"]))
    with open(os.path.join(GOLDEN, "ref_sizes.json"), "w") as f:
        json.dump(sizes, f)

    with tempfile.TemporaryDirectory() as tmp:
        # function-level known-answer vectors
        subprocess.check_call([HARNESS, "kat", tmp], stdout=subprocess.DEVNULL)
        widths = {"kat_sphere_hit": 20, "kat_optics": 17, "kat_background": 6, "kat_samplers": 10,
                  "kat_camera": 35, "kat_get_ray": 33, "kat_scatter": 32}
        kats = {}
        for name, w in widths.items():
            a = np.fromfile(os.path.join(tmp, name + ".f32"), dtype=np.float32)
            assert a.size % w == 0, (name, a.size, w)
            kats[name] = a.reshape(-1, w)
        np.savez_compressed(os.path.join(GOLDEN, "ref_kat.npz"), **kats)

        # scene tables (the oracle's generator; the product's generator is tested against these)
        tables = {}
        for key, kw in SCENES.items():
            tables[key] = pyoracle.random_scene(**kw)
            pyoracle.write_scene_bin(os.path.join(tmp, key + ".bin"), tables[key])
        np.savez_compressed(os.path.join(GOLDEN, "scenes.npz"), **tables)
        with open(os.path.join(GOLDEN, "scenes.json"), "w") as f:
            json.dump({k: dict(v, n=int(len(tables[k]))) for k, v in SCENES.items()}, f, indent=1)

        # whole frames through the reference's own functions with the shared stream injected
        meta = {}
        for name, (skey, cols, rows, sb, sc, depth, seed, rec, keep) in FRAMES.items():
            prefix = os.path.join(tmp, name)
            out = subprocess.check_output([HARNESS, "frame", os.path.join(tmp, skey + ".bin"), prefix, str(cols), str(rows),
                                           str(sb), str(sc), str(depth), str(seed), str(rec)], stderr=subprocess.DEVNULL)
            info = json.loads(out.decode().strip().splitlines()[-1])
            arrays = {}
            if "primary" in keep:
                p = np.fromfile(prefix + ".primary.u32", dtype=np.uint32)
                p16 = np.where(p == 0xFFFFFFFF, 0xFFFF, p).astype(np.uint16)   # n < 65535 in every fixture
                arrays["primary_u16"] = p16
            if "accum" in keep:
                arrays["accum"] = np.fromfile(prefix + ".accum.f32", dtype=np.float32)
            if "fb" in keep and sb == 0:
                arrays["fb"] = np.fromfile(prefix + ".fb.u32", dtype=np.uint32)
            np.savez_compressed(os.path.join(GOLDEN, "frame_" + name + ".npz"), **arrays)
            meta[name] = dict(info, scene=skey, cols=cols, rows=rows, spp_begin=sb, spp_count=sc, max_depth=depth, seed=seed,
                              recursive=rec)
            print(name, info)
        with open(os.path.join(GOLDEN, "frames.json"), "w") as f:
            json.dump(meta, f, indent=1)

        # statistical reference: the UNMODIFIED renderScene (thread pool, mt19937 draws) at high spp, small image
        fbp = os.path.join(tmp, "ref.fb.u32")
        out = subprocess.check_output([HARNESS, "render_ref", os.path.join(tmp, "book1.bin"), fbp, "120", "80", "1024", "50", "8", "64", "0", "99"],
                                      stderr=subprocess.DEVNULL)
        info = json.loads(out.decode().strip().splitlines()[-1])
        np.savez_compressed(os.path.join(GOLDEN, "refrender_book1_120x80_spp1024.npz"), fb=np.fromfile(fbp, dtype=np.uint32))
        with open(os.path.join(GOLDEN, "refrender.json"), "w") as f:
            json.dump({"book1_120x80_spp1024": dict(scene="book1", cols=120, rows=80, spp=1024, max_depth=50, **info)}, f, indent=1)
        print("render_ref", info)


if __name__ == "__main__":
    main()
