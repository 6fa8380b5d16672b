"""A/B timing of librtx builds on the bench scene: alternates the libraries (one subprocess per measurement, so every run
starts from the same state) and prints min / median trace-kernel ms per library.
    python tools/ab.py <spp> <reps> libA.so libB.so [...]"""
import os, subprocess, sys, statistics
spp, reps, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {l: [] for l in libs}
for r in range(reps):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["RTX_LIB_PATH"] = l
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "quick.py"), spp], env=env, capture_output=True, text=True, timeout=300).stdout
        ms = float(out.split("trace_ms")[1].split()[0])
        res[l].append(ms)
for l in libs:
    v = res[l]
    print("%-40s min %.2f  median %.2f  all %s" % (l, min(v), statistics.median(v), " ".join("%.2f" % x for x in v)))
