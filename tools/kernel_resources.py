#!/usr/bin/env python3
"""Registers, spills, LDS and occupancy of every kernel of liblpf, from the compiler's own report
(-Rpass-analysis=kernel-resource-usage).  usage: python tools/kernel_resources.py [name filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-c",
           os.path.join(ROOT, "lidar_object_detection_amd/csrc/lpf_api.hip"), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    for name, r in rows.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt and flt not in dem:
            continue
        print("%-70s VGPR %3s  AGPR %2s  SGPR %3s  spill %s/%s  scratch %s  LDS %6s  occ %s" % (
            dem[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
            r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
