"""Builds liblpf.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so
travels to the GPU box with the repository snapshot.
"""
import hashlib
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


HEADER = os.path.join(os.path.dirname(_HERE), "include", "lpf.h")
_MARKER = b"LPF_BUILD_ID="


def source_id(lab=False):
    """What a library built NOW would be built from: the first 16 hex digits of the SHA-256 over the sources of csrc/, include/lpf.h and
    the compiler flags (-DLPF_LAB for the lab build).  It is compiled into the library (lpf_build_id()), so a binary can always be tied
    to its sources: needs_build() and the loader compare the two, bench.py prints it with every number."""
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]:
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(FLAGS + (["-DLPF_LAB"] if lab else [])).encode())
    return h.hexdigest()[:16]


def library_id(lib):
    """The build id a library file carries (read from the file, nothing is loaded), or None: no file / a build without one."""
    try:
        with open(lib, "rb") as f:
            data = f.read()
    except OSError:
        return None
    i = data.find(_MARKER)
    if i < 0:
        return None
    tail = data[i + len(_MARKER):i + len(_MARKER) + 16]
    return tail.decode("ascii", "replace") if len(tail) == 16 and all(c in b"0123456789abcdef" for c in tail) else None


def needs_build(lib=None):
    """True unless `lib` was built from exactly the sources (and flags) that are here now.  Ids, not mtimes: a checkout, a copy to the
    GPU box or a touched file changes the mtimes but not what the library is."""
    lib = lib or LIB
    return library_id(lib) != source_id(lab=os.path.abspath(lib) == os.path.abspath(LAB_LIB))


def torch_lib_dir():
    """Directory of the HIP runtime the PyTorch-ROCm wheel bundles (libamdhip64.so), or None."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        tl = os.path.join(os.path.dirname(spec.origin), "lib") if spec and spec.origin else None
    except Exception:
        tl = None
    return tl if tl and os.path.exists(os.path.join(tl, "libamdhip64.so")) else None


def build(force=False, verbose=False, lab=False):
    """liblpf.so (the product), or with lab=True liblpf_lab.so: the same sources compiled with -DLPF_LAB.

    A process must hold ONE HIP runtime.  The PyTorch-ROCm wheel bundles its own copy and its libraries ask for it by the
    name ``libamdhip64.so``; /opt/rocm's carries the SONAME ``libamdhip64.so.7``, which is what a plain hipcc link records.
    The loader takes those for two different libraries: a liblpf.so linked the plain way and loaded BEFORE torch left the
    process with both runtimes, and torch then failed with "No HIP GPUs are available".  So the library is linked in two
    steps: compile, then link against a stub that carries the SONAME ``libamdhip64.so`` (only the name is recorded), with an
    RPATH of torch's lib directory (if torch is installed) and /opt/rocm/lib.  Whichever of liblpf.so and torch is loaded
    first, the other finds the runtime already mapped under the name it asks for."""
    lib = LAB_LIB if lab else LIB
    if not force and not needs_build(lib):
        return lib
    work = os.path.join(_HERE, "build", "lab" if lab else "lib")
    os.makedirs(os.path.join(work, "stub"), exist_ok=True)
    obj = os.path.join(work, "lpf_api.o")
    compile_flags = [f for f in FLAGS if f not in ("-shared", "-pthread", "-ldl")]
    cmds = [[hipcc()] + compile_flags + (["-DLPF_LAB"] if lab else []) + ['-DLPF_BUILD_ID_STR="%s"' % source_id(lab),
                                                                          "-c", os.path.join(CSRC, "lpf_api.hip"), "-o", obj]]
    if verbose:
        print(" ".join(cmds[0]))
    subprocess.check_call(cmds[0], cwd=CSRC)
    undefined = subprocess.run(["nm", "-u", obj], check=True, capture_output=True, text=True).stdout.split()
    names = sorted({n for n in undefined if n.startswith(("hip", "__hip"))})
    stub_c = os.path.join(work, "stub", "stub.c")
    with open(stub_c, "w") as f:
        f.write("/* link-time stand-in for the HIP runtime: only its SONAME is recorded in liblpf.so */\n")
        f.write("".join("void %s(void) {}\n" % n for n in names))
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-Wl,-soname,libamdhip64.so", stub_c, "-o", os.path.join(work, "stub", "libamdhip64.so")])
    rpath = ":".join([d for d in (torch_lib_dir(), "/opt/rocm/lib") if d])
    link = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-no-hip-rt", obj, "-L" + os.path.join(work, "stub"), "-lamdhip64",
            "-Wl,-rpath," + rpath, "-Wl,--disable-new-dtags", "-pthread", "-ldl", "-o", lib]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, lab="lab" in sys.argv[1:]))
