"""The standalone C++ host (rtx_cli: the reference's flags + the HIP mode) end to end: PPM on disk equals the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cli_writes_reference_format_ppm(tmp_path, oracle, golden_scenes):
    cli = os.path.join(ROOT, "raytracing_amd", "rtx_cli")
    if not os.path.exists(cli):
        pytest.fail("raytracing_amd/rtx_cli is not built (make -C raytracing_amd/csrc cli)")
    out = tmp_path / "img.ppm"
    cols, rows, spp, depth = 60, 40, 3, 7
    r = subprocess.run([cli, "-w", str(cols), "-h", str(rows), "-a", str(spp), "-m", str(depth), "-s", "1", "-f", str(out), "-t", "4", "-b", "16"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = open(out).read().split("\n")
    assert lines[:3] == ["P3", "%d %d" % (cols, rows), "255"]
    px = np.array([[int(v) for v in ln.split()] for ln in lines[3:3 + cols * rows]], dtype=np.uint32)
    accum, _, _ = oracle.render(golden_scenes["book1"], oracle.book_camera(cols, rows), rows, cols, 0, spp, depth, 1)
    fb = oracle.resolve(accum, rows, cols, spp)
    want = np.stack([(fb >> 24) & 0xFF, (fb >> 16) & 0xFF, (fb >> 8) & 0xFF], axis=1)
    assert (px == want).all()
    # -c / -i name back ends this build does not have
    r = subprocess.run([cli, "-c"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "not available" in r.stdout


def test_cli_progressive_resume(tmp_path):
    """-p / -k / -R / -6: a render stopped after some passes and resumed from its checkpoint ends with the same image file
    as the uninterrupted progressive render."""
    cli = os.path.join(ROOT, "raytracing_amd", "rtx_cli")
    base = [cli, "-w", "64", "-h", "40", "-m", "6", "-s", "3", "-p", "2", "-6"]
    a, b, ck = tmp_path / "a.ppm", tmp_path / "b.ppm", tmp_path / "b.accum"
    r = subprocess.run(base + ["-a", "6", "-f", str(a)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.count("pass done") == 3, r.stdout + r.stderr
    r = subprocess.run(base + ["-a", "4", "-f", str(b), "-k", str(ck)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and ck.exists(), r.stdout + r.stderr
    r = subprocess.run(base + ["-a", "6", "-f", str(b), "-k", str(ck), "-R"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "resuming" in r.stdout and r.stdout.count("pass done") == 1, r.stdout + r.stderr
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a, "rb").read().startswith(b"P6\n64 40\n255\n")


def test_render_dist_matches_rtx_cli_and_resumes(tmp_path):
    """python -m raytracing_amd.render_dist (the multi-GPU host; one rank here): same image file as rtx_cli for the same
    passes, checkpoints written by one are resumed by the other, a finished job resumed only rewrites its image."""
    import sys
    cli = os.path.join(ROOT, "raytracing_amd", "rtx_cli")
    common = ["-w", "64", "-h", "40", "-m", "6", "-s", "3", "-p", "2", "-6", "-r"]
    dist_cmd = [sys.executable, "-m", "raytracing_amd.render_dist"] + common
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    a, b, ck = tmp_path / "a.ppm", tmp_path / "b.ppm", tmp_path / "b.accum"
    r = subprocess.run([cli] + common + ["-a", "6", "-f", str(a)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(dist_cmd + ["-a", "4", "-f", str(b), "-k", str(ck), "-F", "auto"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.count("pass done") == 2 and ck.exists(), r.stdout + r.stderr
    # resumed by the C++ host: the checkpoint format is shared
    r = subprocess.run([cli] + common + ["-a", "6", "-f", str(b), "-k", str(ck), "-R"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "resuming" in r.stdout and r.stdout.count("pass done") == 1, r.stdout + r.stderr
    assert open(a, "rb").read() == open(b, "rb").read()
    # ... and the finished job resumed by the Python host renders nothing and writes the same image
    os.remove(b)
    r = subprocess.run(dist_cmd + ["-a", "6", "-f", str(b), "-k", str(ck), "-R"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and "resuming" in r.stdout and "at 6 of 6" in r.stdout, r.stdout + r.stderr
    assert open(a, "rb").read() == open(b, "rb").read()
    # a checkpoint that is further along than the job asked for is not continued
    r = subprocess.run(dist_cmd + ["-a", "2", "-f", str(b), "-k", str(ck), "-R"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and "starting over" in r.stdout and r.stdout.count("pass done") == 1, r.stdout + r.stderr
    r = subprocess.run([cli] + common + ["-a", "2", "-f", str(a)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and open(a, "rb").read() == open(b, "rb").read()


def test_cli_filter_flag(tmp_path):
    """-F valu|mfma|grid|auto: the same image file whichever way the candidates are found; an unknown name is refused."""
    cli = os.path.join(ROOT, "raytracing_amd", "rtx_cli")
    base = [cli, "-w", "64", "-h", "40", "-m", "8", "-s", "5", "-a", "3", "-6", "-n", "scaled"]
    images = []
    for f in ("valu", "mfma", "grid", "auto"):
        out = tmp_path / (f + ".ppm")
        r = subprocess.run(base + ["-F", f, "-f", str(out)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        images.append(open(out, "rb").read())
    assert all(img == images[0] for img in images[1:])
    r = subprocess.run(base + ["-F", "bvh", "-f", str(tmp_path / "x.ppm")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "-F takes" in r.stdout


def test_two_ranks_share_the_gpu_in_rehearsal(tmp_path):
    """The one-process-per-GPU hosts with TWO ranks on the one GPU of a test box (RTX_DIST_BACKEND=gloo: the sum goes through
    the host instead of RCCL, everything else is the N > 1 code): render_dist's image equals the one-rank image up to the
    order of the float sums (at most one 8-bit step in a few channels), and bench.py prints its line for --gpus 2."""
    import json
    import sys
    env = dict(os.environ, PYTHONPATH=ROOT, RTX_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    common = ["-w", "96", "-h", "64", "-m", "8", "-s", "2", "-a", "12", "-p", "4", "-6"]
    a, b = tmp_path / "one.ppm", tmp_path / "two.ppm"
    r = subprocess.run([sys.executable, "-m", "raytracing_amd.render_dist"] + common + ["-f", str(a)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29617"]
    r = subprocess.run(launch + ["-m", "raytracing_amd.render_dist"] + common + ["-f", str(b)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    pa, pb = np.frombuffer(open(a, "rb").read()[-96 * 64 * 3:], dtype=np.uint8).astype(int), np.frombuffer(open(b, "rb").read()[-96 * 64 * 3:], dtype=np.uint8).astype(int)
    assert np.abs(pa - pb).max() <= 1 and (pa != pb).mean() < 0.01
    # bench.py --gpus 2 under a launcher: strong scaling is the headline (the image's 9 samples split 5 + 4 over the ranks:
    # BASELINE's 500 spp over 8 GPUs is timed this way), the weak figure (9 spp per rank) rides along
    r = subprocess.run(launch + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "9", "--cols", "160", "--rows", "100"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and "REHEARSAL" in line["config"]["parallelism"] and "9 spp in all" in line["config"]["workload"]
    assert line["value"] > 0 and "cpu_baseline" not in line and "other_workloads" not in line
    assert line["rccl_ranks"] == 2 and len(line["member_trace_ms"]) == 2 and line["exchange"]["allreduce_ms"] > 0
    assert abs(line["value"] * line["ms_per_step"] / (160 * 100 * 9 / 1e3) - 1.0) < 1e-3
    weak = line["weak_scaling"]
    assert abs(weak["value"] * weak["ms_per_step"] / (160 * 100 * 18 / 1e3) - 1.0) < 1e-3
    # --weak: the old headline
    r = subprocess.run(launch + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "8", "--cols", "160", "--rows", "100", "--weak"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "weak" and "8 spp per GPU" in line["config"]["workload"] and "strong_scaling" in line
    assert abs(line["value"] * line["ms_per_step"] / (160 * 100 * 16 / 1e3) - 1.0) < 1e-3


def test_bench_drives_a_group_without_a_launcher():
    """`python bench.py --gpus 2` with no launcher (WORLD_SIZE unset): ONE process drives the devices through rtx_group_* --
    rehearsed here with two members on the one GPU (RTX_GROUP_LOOPBACK=1; the line says REHEARSAL).  Strong scaling is the
    headline, the weak figure, the members' trace-kernel times and the exchange's own time ride along."""
    import json
    import sys
    env = dict(os.environ, RTX_GROUP_LOOPBACK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RTX_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "9", "--cols", "160", "--rows", "100"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                  # stdout carries the JSON line and nothing else
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "strong" and "REHEARSAL" in line["config"]["parallelism"]
    assert "9 spp in all" in line["config"]["workload"] and len(line["member_trace_ms"]) == 2 and all(m > 0 for m in line["member_trace_ms"])
    assert abs(line["value"] * line["ms_per_step"] / (160 * 100 * 9 / 1e3) - 1.0) < 1e-3
    assert abs(line["weak_scaling"]["value"] * line["weak_scaling"]["ms_per_step"] / (160 * 100 * 18 / 1e3) - 1.0) < 1e-3
    assert line["exchange"]["allreduce_ms"] > 0 and line["roofline"]["frac"] > 0
    assert 2.0 < line["config"]["rays_per_sample"] < 3.5
