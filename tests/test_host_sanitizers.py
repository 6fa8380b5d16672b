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
