"""Sanitizers for the code that runs on the host ("race detection / sanitizers"; build container only -- never on a
GPU box, where sanitizer runs are not available):

  * oracle/lpf_oracle.c under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`), running the oracle's own
    golden-vector and mask tests against the sanitized library;
  * the HOST side of liblpf -- lpf_api.hip compiled `--offload-host-only` and linked against tests/host_san/fake_hip.cpp, a functional
    stand-in for the HIP runtime (device memory = heap blocks, copies = memcpy, launches = nothing) -- under ASan + UBSan and under
    ThreadSanitizer, driven by tests/host_san/drive.cpp: argument validation, host-memory runs, > 24 000 software-pipelined runs with
    new masks / rectangles / boxes / batch shapes (the pinned upload ring is lapped many times, with pieces that fill a quarter
    exactly: ADVICE round 3), the box-set and scratch-set rotation, the graph state machine, and the reader's worker threads.
"""
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _gcc_lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_gcc_lib("libasan.so") is None, reason="gcc's libasan is not installed")
def test_oracle_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=_gcc_lib("libasan.so"), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               LPF_ORACLE_SO=os.path.join(REPO, "oracle", "liblpf_oracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_oracle_golden.py", "tests/test_oracle_masks.py", "-q", "-x", "-p", "no:cacheprovider"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    text = r.stdout + r.stderr
    assert r.returncode == 0 and " passed" in text, text[-3000:]
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-3000:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc is not installed")
@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_host_side_of_the_library_under_sanitizers(tmp_path, san):
    out = str(tmp_path / "build")
    b = subprocess.run(["make", "-C", os.path.join(REPO, "tests", "host_san"), san, "OUT=" + out, "HIPCC=" + HIPCC], capture_output=True, text=True, timeout=900)
    assert b.returncode == 0, (b.stdout + b.stderr)[-3000:]
    scans = tmp_path / "scans"
    scans.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(out, "drive_" + san), str(scans), "12000"], capture_output=True, text=True, timeout=900, env=env)
    text = r.stdout + r.stderr
    assert r.returncode == 0 and "drive: 0 failed checks" in text, text[-4000:]
    assert "Sanitizer" not in text and "runtime error" not in text, text[-4000:]
