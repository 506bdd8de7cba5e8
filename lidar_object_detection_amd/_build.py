"""Builds liblpf.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so
travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "liblpf.so")
LAB_LIB = os.path.join(_HERE, "liblpf_lab.so")       # the same sources with -DLPF_LAB: lpf_set_geometry (tools/, forced-geometry tests)
SOURCES = ("lpf_api.hip", "lpf_kernels.hip.h", "lpf_reader.hip.h")
ARCH = "gfx950"

# -ffp-contract=off: the kernels spell out every fma() the reference's BLAS performs;
# the compiler must not fuse (or split) anything else, or integer outputs can flip.
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ldl",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")
    return exe


def needs_build(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(_HERE), "include", "lpf.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, lab=False):
    """liblpf.so (the product), or with lab=True liblpf_lab.so: the same sources compiled with -DLPF_LAB."""
    lib = LAB_LIB if lab else LIB
    if not force and not needs_build(lib):
        return lib
    cmd = [hipcc()] + FLAGS + (["-DLPF_LAB"] if lab else []) + ["-o", lib, os.path.join(CSRC, "lpf_api.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, lab="lab" in sys.argv[1:]))
