#!/usr/bin/env python3
"""LAB: build an alternative liblpf.so for a same-box A/B -- the product's sources with extra compiler flags -- into
lidar_object_detection_amd/build/alt/<name>.so (git-ignored; it travels to the GPU box, so delete it when the experiment is over).
Select it with `PROBE_LIBRARY=<path> tools/real_probe.py ...`, `LpfContext(0, library=<path>)` or tools/lab_so_sweep.sh.
usage: python tools/build_alt.py <name> [-DMACRO=value ...]      e.g.  build_alt.py spread11 -DLPF_TAIL_SPREAD_20THS=11"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidar_object_detection_amd import _build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(os.path.dirname(b.LIB), "build", "alt")
os.makedirs(out, exist_ok=True)
b.FLAGS.extend(flags)
b.LAB_LIB = os.path.join(out, "liblpf_%s.so" % name)
print(b.build(force=True, lab=True, verbose=True))
