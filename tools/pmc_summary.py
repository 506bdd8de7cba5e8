#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one counter per pass) into
per-kernel HBM traffic per launch, with the gfx950 corrections of MI355X_MICROARCH.md (HBM):
  * both counters are in KiB;
  * FETCH_SIZE reports exactly half the bytes of a wide coalesced (16 B/lane) streaming read
    -> doubled (calibrated here on tools/k1_lab's ref_read16: 32 MB stream reads 15636 KiB);
  * WRITE_SIZE is exact for the 16-, 8- and 4-byte-per-lane streaming stores used here
    (calibrated on ref_copy16 and ref_copy_16_8_4: 32.0 MB and 24.0 MB).
The gather part of a kernel's reads is doubled along with the stream, so the read figure is an
upper bound.

usage: pmc_summary.py <dir with <prefix>_FETCH_SIZE and <prefix>_WRITE_SIZE subdirs> <prefix> [<prefix2> ...] <out.json>
The output records, under "_meta", the sha of the kernel sources the passes ran (bench.py refuses a figure from other sources).
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    root, prefixes, out = sys.argv[1], sys.argv[2:-1], sys.argv[-1]
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)       # a directory may hold earlier passes too
    f, w = {}, {}
    for prefix in prefixes:
        f.update(per_kernel(newest("%s/%s_FETCH_SIZE/*/*_counter_collection.csv" % (root, prefix))))
        w.update(per_kernel(newest("%s/%s_WRITE_SIZE/*/*_counter_collection.csv" % (root, prefix))))
    res = {"_meta": {"kernel_source_sha16": bench.kernel_source_sha(), "points_per_launch": 8 * bench.N_POINTS,
                     "passes": prefixes, "corrections": "FETCH_SIZE KiB x 2 (gfx950, wide coalesced reads), WRITE_SIZE KiB exact"}}
    for k in sorted(set(f) | set(w)):
        if "lpf" not in k and "ref_" not in k:
            continue
        fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
        res[k] = {"launches": f.get(k, (0, 0))[1], "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                  "read_bytes_corrected": 2.0 * fk * 1024.0, "write_bytes": wk * 1024.0,
                  "hbm_bytes_per_launch": 2.0 * fk * 1024.0 + wk * 1024.0}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k == "_meta":
            continue
        print("%-60s %10.1f MB read  %10.1f MB written" % (k[-60:], v["read_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
