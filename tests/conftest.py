import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def golden_scenes():
    z = np.load(os.path.join(GOLDEN, "scenes.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_frames():
    with open(os.path.join(GOLDEN, "frames.json")) as f:
        meta = json.load(f)
    out = {}
    for name, m in meta.items():
        z = np.load(os.path.join(GOLDEN, "frame_" + name + ".npz"))
        d = dict(m)
        for k in z.files:
            d[k] = z[k]
        if "primary_u16" in d:
            p = d["primary_u16"].astype(np.uint32)
            d["primary"] = np.where(p == 0xFFFF, np.uint32(0xFFFFFFFF), p).astype(np.uint32)
        out[name] = d
    return out


@pytest.fixture(scope="session")
def golden_kat():
    z = np.load(os.path.join(GOLDEN, "ref_kat.npz"))
    return {k: z[k] for k in z.files}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
