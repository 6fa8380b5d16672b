#!/usr/bin/env python3
"""Adds statistical reference images to tests/golden/: the UNMODIFIED renderScene of the reference (thread pool, its own
mt19937 draws; oracle/_ref/ref_harness render_ref) at 1024 spp on small frames, for the scenes / modes that
oracle/gen_fixtures.py does not cover (glass-heavy iterative, book-1 with -r, the 1601-sphere scaled scene).  Runs here only (needs /root/reference
compiled by `make -C oracle ref`); the outputs are data fixtures.  Existing entries are kept: the reference's stream is
racy, so a re-run would not reproduce them bit for bit."""
import json, os, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pyoracle
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
HARNESS = os.path.join(HERE, "_ref", "ref_harness")
JOBS = {"glass_120x80_spp1024": ("glass", 0), "book1_120x80_spp1024_recursive": ("book1", 1), "scaled_120x80_spp1024": ("scaled", 0)}

def main():
    scenes = np.load(os.path.join(GOLDEN, "scenes.npz"))
    with open(os.path.join(GOLDEN, "refrender.json")) as f:
        meta = json.load(f)
    with tempfile.TemporaryDirectory() as tmp:
        for name, (skey, rec) in JOBS.items():
            if name in meta:
                continue
            pyoracle.write_scene_bin(os.path.join(tmp, skey + ".bin"), scenes[skey])
            fbp = os.path.join(tmp, "ref.fb.u32")
            out = subprocess.check_output([HARNESS, "render_ref", os.path.join(tmp, skey + ".bin"), fbp, "120", "80", "1024", "50", "8", "64", str(rec), "99"],
                                          stderr=subprocess.DEVNULL)
            info = json.loads(out.decode().strip().splitlines()[-1])
            np.savez_compressed(os.path.join(GOLDEN, "refrender_" + name + ".npz"), fb=np.fromfile(fbp, dtype=np.uint32))
            meta[name] = dict(scene=skey, cols=120, rows=80, spp=1024, max_depth=50, recursive=rec, **info)
            print(name, info, flush=True)
    with open(os.path.join(GOLDEN, "refrender.json"), "w") as f:
        json.dump(meta, f, indent=1)

if __name__ == "__main__":
    main()
