import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(frame):
    return dict(np.load(os.path.join(GOLDEN, "frame_%010d.npz" % frame)))


def load_calib():
    return dict(np.load(os.path.join(GOLDEN, "calib_cam0.npz")))


def golden_frames():
    import json
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)


def unpack_masks(g, kind, H, W):
    """float32 [M,H,W] 0/1 masks from the bit-packed golden copy."""
    packed = g["masks_%s_packed" % kind]
    return np.unpackbits(packed, axis=-1)[..., :W].astype(np.float32)


@pytest.fixture(scope="session")
def calib():
    return load_calib()
