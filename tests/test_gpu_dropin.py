"""The drop-in wiring end to end on the GPU: raytracing_amd/dropin_twin is raytracing_amd/host/dropin_demo.cpp -- a main()
around renderSceneHIP, written as INTEGRATION.md section 2 tells a maintainer to wire the fourth backend in -- built against
host/scene_types.h, the product's own Scene / Sphere / material_t / Camera with the reference's names and fields.  Its
framebuffer must equal, bit for bit, the golden frame that the reference's own _renderJob produced from the same scene.
The build of the SAME source against the reference's own headers (oracle/_ref/dropin_demo) contains reference object code
and therefore stays in the container that holds the reference: tests/test_host_cabi.py checks there that it compiles, links
and fails loudly without a GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEMO = os.path.join(ROOT, "raytracing_amd", "dropin_twin")


@pytest.mark.parametrize("name", ["book1_96x64_spp2", "book1_96x64_spp2_recursive", "glass_96x64_spp2", "book1_96x64_spp2_depth3"])
def test_scene_types_through_render_scene_hip(tmp_path, oracle, golden_scenes, golden_frames, name):
    assert os.path.exists(DEMO), "build raytracing_amd/dropin_twin first: make -C raytracing_amd/csrc cli"
    f = golden_frames[name]
    assert f["spp_begin"] == 0 and f["seed"] == 1 and "fb" in f
    scene_bin, out = tmp_path / "scene.bin", tmp_path / "out.fb.u32"
    oracle.write_scene_bin(str(scene_bin), golden_scenes[f["scene"]])
    r = subprocess.run([DEMO, str(scene_bin), str(out), str(f["cols"]), str(f["rows"]), str(f["spp_count"]), str(f["max_depth"]), str(f["recursive"])],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    fb = np.fromfile(out, dtype=np.uint32)
    assert (fb == f["fb"]).all(), name
