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
    assert lib.lpf_abi_version() == 8
    # the measured-slower machinery of ABI 4 is gone from the product, and the geometry override lives in the lab build only
    assert not hasattr(lib, "lpf_set_cu_partition") and not hasattr(lib, "lpf_set_geometry") and not hasattr(lib, "lpf_lab_role_clock")
    assert _declared(lab=True) == sorted(names + ["lpf_set_geometry", "lpf_lab_role_clock"])


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


_ORDER_SNIPPET = """
import ctypes, sys
{first}
{second}
maps = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
print("RUNTIMES", len(maps), maps)
{gpu}
"""


def _run_order(first, second, gpu=""):
    import subprocess
    import sys
    lib = _native.library_path()
    code = _ORDER_SNIPPET.format(first=first.format(lib=lib), second=second.format(lib=lib), gpu=gpu)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=REPO, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_one_hip_runtime_whichever_is_loaded_first():
    """liblpf.so and PyTorch-ROCm in one process, in both load orders, through raw ctypes (no Python wrapper to put them in
    order): exactly one libamdhip64 may end up mapped.  (ABI <= 4 was linked against /opt/rocm's SONAME and, loaded first, left
    the process with two runtimes -- torch then reported "No HIP GPUs are available".)"""
    _build.build()
    load_lpf, load_torch = 'lib = ctypes.CDLL("{lib}")', "import torch"
    for first, second in ((load_lpf, load_torch), (load_torch, load_lpf)):
        out = _run_order(first, second)
        assert "RUNTIMES 1 " in out, out


def test_binary_is_tied_to_its_sources(monkeypatch, tmp_path):
    """The library carries the id of the sources + flags it was compiled from (lpf_build_id); staleness is decided by that id, not by
    mtimes; a stale library of the package is rebuilt at load, or refused when it cannot be -- never loaded as it is."""
    _build.build()
    sid = _build.source_id()
    assert len(sid) == 16 and _build.library_id(_build.LIB) == sid and not _build.needs_build(_build.LIB)
    lib = _native.load()
    assert lib._lpf_info["build_id"] == sid and lib._lpf_info["path"] == os.path.abspath(_build.LIB)
    os.utime(os.path.join(_build.CSRC, "lpf_api.hip"))                      # a touched source is not a changed source
    assert not _build.needs_build(_build.LIB)
    assert _build.source_id(lab=True) != sid                                 # the flags are part of the id
    junk = tmp_path / "liblpf.so"
    junk.write_bytes(b"\x7fELF" + b"\0" * 64)
    assert _build.library_id(str(junk)) is None and _build.library_id(str(tmp_path / "absent.so")) is None
    # the sources "change": the library on disk is now stale
    monkeypatch.setattr(_build, "source_id", lambda lab=False: "0123456789abcdef")
    assert _build.needs_build(_build.LIB)
    monkeypatch.setattr(_native, "_libs", {})
    calls = []

    def cannot(**kw):
        calls.append(kw)
        raise RuntimeError("hipcc not found")
    monkeypatch.setattr(_build, "build", cannot)
    try:
        _native.load()
    except _native.LpfError as e:
        assert "stale" in str(e) and sid in str(e) and "0123456789abcdef" in str(e)
    else:
        raise AssertionError("a stale liblpf.so was loaded")
    assert calls == [{"lab": False}]                                         # ... after an attempt to rebuild it
    # a rebuild that does not produce the expected id is refused as well (the loaded binary says what it is)
    monkeypatch.setattr(_build, "build", lambda **kw: calls.append(kw))
    try:
        _native.load()
    except _native.LpfError as e:
        assert "build id" in str(e)
    else:
        raise AssertionError("a library with another build id was accepted")


def test_lpf_library_needs_a_lab_run(monkeypatch):
    monkeypatch.setenv("LPF_LIBRARY", _build.LAB_LIB)
    monkeypatch.delenv("LPF_LAB", raising=False)
    try:
        _native.library_path()
    except _native.LpfError as e:
        assert "lab run" in str(e)
    else:
        raise AssertionError("LPF_LIBRARY swapped the library outside a lab run")
    monkeypatch.setenv("LPF_LAB", "1")
    assert _native.library_path() == _build.LAB_LIB
