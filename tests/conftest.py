import os
import sys

import numpy as np
import pytest

try:                     # torch's bundled HIP runtime has to be the first one loaded in a process that uses both
    import torch  # noqa: F401
except Exception:
    pass

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


def lab_library():
    """liblpf_lab.so: the product's sources compiled with -DLPF_LAB, which adds lpf_set_geometry (the product picks the launch
    geometry by launch size only).  Built by __graft_entry__.build(); the forced-geometry tests skip without it."""
    from lidar_object_detection_amd import _build
    return _build.LAB_LIB if os.path.exists(_build.LAB_LIB) else None


FORMS = ["auto", "small", "small-narrow", "small-1024", "large", "large-scan"]


def context_for_form(form, device=0):
    """An LpfContext of the product library ("auto": geometry by launch size, what ships) or of the lab build with the launch
    geometry forced, so that small inputs reach the code paths of large launches."""
    from lidar_object_detection_amd._native import LpfContext
    if form == "auto":
        return LpfContext(device)
    lab = lab_library()
    if lab is None:
        pytest.skip("liblpf_lab.so has not been built (python -m lidar_object_detection_amd._build lab)")
    c = LpfContext(device, library=lab)
    c.set_geometry(form)
    return c


def load_golden_full(frame):
    """Full-size variant (tests/golden/make_golden.py, HASH_FRAMES): inputs in full, long outputs as SHA-256 digests."""
    return dict(np.load(os.path.join(GOLDEN, "frame_%010d_full.npz" % frame)))


def check_full(g, key, arr, dtype):
    """arr == the reference's array ``key`` of a full-size golden: compared directly when it was stored, by length and
    SHA-256 of its ``dtype`` bytes when only the digest was."""
    import hashlib
    a = np.ascontiguousarray(arr, dtype=dtype)
    if key in g:
        assert np.array_equal(a, g[key]), key
    else:
        assert a.size == int(g[key + "_len"]), (key, a.size, int(g[key + "_len"]))
        assert hashlib.sha256(a.tobytes()).digest() == g[key + "_sha256"].tobytes(), key
