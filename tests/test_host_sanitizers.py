"""rtx_host.cpp (grid builder, chunk planner, scene generator, camera: the library's host arithmetic) compiled for the CPU with
AddressSanitizer and UBSan and driven by tests/native/host_fuzz.cpp on random and degenerate inputs.  No GPU, no HIP runtime."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_host_code_under_asan_and_ubsan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / "host_fuzz"
    src = [os.path.join(ROOT, "tests", "native", "host_fuzz.cpp"), os.path.join(ROOT, "raytracing_amd", "csrc", "rtx_host.cpp")]
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", os.path.join(ROOT, "include"), "-o", str(exe)] + src, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "host fuzz ok" in r.stdout, (r.stdout + r.stderr)[-4000:]


def test_oracle_tile_pool_under_thread_sanitizer(tmp_path):
    """SURVEY s5 "race detection": the one threaded piece of host code in the repository -- the oracle's tile pool, which plays
    the role of the reference's thread_pool.cpp workers -- built with -fsanitize=thread and run on 2, 5 and 8 threads; the frames
    must equal the one-thread frame and ThreadSanitizer must stay silent."""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = tmp_path / "oracle_tsan"
    src = [os.path.join(ROOT, "tests", "native", "oracle_tsan.c"), os.path.join(ROOT, "oracle", "rt_oracle.c")]
    r = subprocess.run(["gcc", "-std=c11", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off", "-o", str(exe)] + src + ["-lm", "-lpthread"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66"))
    if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel")
    assert r.returncode == 0 and "oracle tsan ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout + r.stderr)[-4000:]
