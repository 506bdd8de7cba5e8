#!/usr/bin/env python3
"""Summarise the counter passes of tools/pmc_real.sh: per run (real_serial, real_fused-pack, synth_serial), per kernel and per
counter the mean value per launch over the steady launches, plus the derived figures the question needs -- HBM bytes per launch
(FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE exact; both in KiB), the L2 hit rate
TCC_HIT / (TCC_HIT + TCC_MISS), L1 -> L2 requests per point.  "_meta" records the sha of the kernel sources that ran.
usage: pmc_real_summary.py <gpurun_out/pmc_real> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "lpf" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    root, out = sys.argv[1], sys.argv[2]
    import bench
    res = {"_meta": {"kernel_source_sha16": bench.kernel_source_sha(),
                     "corrections": "FETCH_SIZE KiB x 2 (gfx950, wide coalesced reads), WRITE_SIZE KiB exact; every counter is the mean per launch",
                     "real": "tools/real_probe.py: 146 real frames (the four full-size golden frames in turn), 16.9 M points, 5 masks each, per step",
                     "synth": "bench.py --mode serial: 8 x 2 M-point clouds, 8 masks, 32 boxes each, per step"}}
    for d in sorted(glob.glob(os.path.join(root, "pmc_*_g*"))):
        if not os.path.isdir(d):
            continue
        run = os.path.basename(d)[4:].rsplit("_g", 1)[0]
        files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = per_kernel(max(files, key=os.path.getmtime))
        for k, cs in agg.items():
            e = res.setdefault(run, {}).setdefault(k, {})
            for c, v in cs.items():
                # (the first launches of a run fill the pipeline / warm the caches: the median launch describes the steady state)
                v = sorted(v)
                e[c] = v[len(v) // 2]
                e["launches"] = len(v)
    for run, ks in res.items():
        if run == "_meta":
            continue
        for k, e in ks.items():
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                e["hbm_read_MB"] = round(2.0 * e["FETCH_SIZE"] * 1024 / 1e6, 2)
                e["hbm_write_MB"] = round(e["WRITE_SIZE"] * 1024 / 1e6, 2)
            if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
                e["l2_hit_rate"] = round(e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
            if "SQ_WAVE_CYCLES" in e and e.get("SQ_WAVE_CYCLES"):
                e["wait_any_share"] = round(e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 4)
                e["active_share"] = round(e.get("SQ_ACTIVE_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 4)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for run, ks in res.items():
        if run == "_meta":
            continue
        for k, e in ks.items():
            print("%-18s %-46s %s" % (run, k[:46], " ".join("%s=%s" % (a, e[a]) for a in ("hbm_read_MB", "hbm_write_MB", "l2_hit_rate", "TCP_TCC_READ_REQ_sum", "wait_any_share") if a in e)))


if __name__ == "__main__":
    main()
