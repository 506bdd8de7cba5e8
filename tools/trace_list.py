#!/usr/bin/env python3
"""The last N lpf_* kernels of a rocprofv3 --kernel-trace CSV, one line each: start and end (us from the first listed one),
duration, gap to the previous end, grid size.  usage: trace_list.py <kernel_trace.csv> [N]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lpf_" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
rows = rows[-n:]
t0, prev = rows[0]["s"], None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:46]
    grid = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    print("%9.2f %9.2f  dur %7.2f  gap %6.2f  grid %9s  %s" % ((r["s"] - t0) / 1e3, (r["e"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3,
                                                           (r["s"] - prev) / 1e3 if prev else 0.0, grid, name))
    prev = r["e"]
