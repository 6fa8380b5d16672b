"""The drop-in claim end to end: oracle/_ref/dropin_demo is oracle/dropin_demo.cpp (a main() around renderSceneHIP) compiled against THE
REFERENCE'S OWN scene.h / sphere.h / material.h / camera.h / sphere.cpp (oracle/Makefile, target `ref`; only possible where
the reference tree exists, the binary travels with the snapshot).  Its framebuffer must equal, bit for bit, the golden frame
that the reference's own _renderJob produced from the same scene."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEMO = os.path.join(ROOT, "oracle", "_ref", "dropin_demo")


@pytest.mark.parametrize("name", ["book1_96x64_spp2", "book1_96x64_spp2_recursive", "glass_96x64_spp2", "book1_96x64_spp2_depth3"])
def test_reference_types_through_render_scene_hip(tmp_path, oracle, golden_scenes, golden_frames, name):
    if not os.path.exists(DEMO):
        pytest.skip("oracle/_ref/dropin_demo is built only where the reference tree exists (make -C oracle ref)")
    f = golden_frames[name]
    assert f["spp_begin"] == 0 and f["seed"] == 1 and "fb" in f
    scene_bin, out = tmp_path / "scene.bin", tmp_path / "out.fb.u32"
    oracle.write_scene_bin(str(scene_bin), golden_scenes[f["scene"]])
    r = subprocess.run([DEMO, str(scene_bin), str(out), str(f["cols"]), str(f["rows"]), str(f["spp_count"]), str(f["max_depth"]), str(f["recursive"])],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    fb = np.fromfile(out, dtype=np.uint32)
    assert (fb == f["fb"]).all(), name
