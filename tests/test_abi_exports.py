"""liblpf.so loads and exports every entry point include/lpf.h declares (no GPU needed:
nothing is called that touches a device), and the Python-side struct mirrors match the C layout."""
import ctypes
import os
import re

from lidar_object_detection_amd import _build, _native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(lab=False):
    text = open(os.path.join(REPO, "include", "lpf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    if not lab:                         # what only a -DLPF_LAB build declares (and exports)
        text = re.sub(r"#ifdef LPF_LAB.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lpf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    _build.build()                      # hipcc cross-compiles for gfx950 without a GPU (no-op if up to date)
    lib = ctypes.CDLL(_native.library_path())
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_native.EXPORTED) == names
    lib.lpf_abi_version.restype = ctypes.c_int
    assert lib.lpf_abi_version() == 5
    # the measured-slower machinery of ABI 4 is gone from the product, and the geometry override lives in the lab build only
    assert not hasattr(lib, "lpf_set_cu_partition") and not hasattr(lib, "lpf_set_geometry")
    assert _declared(lab=True) == sorted(names + ["lpf_set_geometry"])


def test_struct_mirrors():
    assert ctypes.sizeof(_native.FrameSummary) == 928 == _native.SUMMARY_DTYPE.itemsize
    assert _native.FrameSummary.inst_off.offset == 8 * 34 and _native.FrameSummary.best_box.offset == 8 * 99
    assert ctypes.sizeof(_native.Outputs) == 13 * 8 and _native.Outputs.uv_valid.offset == 11 * 8


def test_no_gpu_means_a_loud_error_not_a_fallback():
    """On a box without a GPU, creating a context must raise; there is no CPU path behind the API."""
    import torch
    if torch.cuda.is_available():
        return
    try:
        _native.LpfContext(0)
    except _native.LpfError as e:
        assert "HIP" in str(e) or "device" in str(e)
    else:
        raise AssertionError("LpfContext() succeeded without a GPU")
