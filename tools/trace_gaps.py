#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: per kernel name the mean duration, and per queue the mean gap between
the end of one kernel and the start of the next (what a step loses between launches).
usage: trace_gaps.py <kernel_trace.csv> [first_kernel_substring]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    key = sys.argv[2] if len(sys.argv) > 2 else "lpf_k1_project"
    rows = [r for r in rows if "lpf_" in r["Kernel_Name"]]
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
    rows.sort(key=lambda r: r["s"])
    # steps: from one K1 start to the next
    k1 = [r for r in rows if key in r["Kernel_Name"]]
    if len(k1) > 20:
        mid = k1[len(k1) // 2:len(k1) // 2 + 40]
        per = [(b["s"] - a["s"]) / 1e3 for a, b in zip(mid, mid[1:])]
        print("K1 start-to-start: mean %.2f us (min %.2f max %.2f) over %d steps" % (sum(per) / len(per), min(per), max(per), len(per)))
        a, b = mid[3], mid[5]
        print("timeline of two steps (us from the first K1 start; queue, kernel, start, end):")
        for r in rows:
            if a["s"] <= r["s"] < b["s"]:
                print("  q%-3s %-30s %8.2f %8.2f  dur %6.2f" % (r.get("Queue_Id", "?"), r["n"], (r["s"] - a["s"]) / 1e3, (r["e"] - a["s"]) / 1e3, (r["e"] - r["s"]) / 1e3))
    dur = collections.defaultdict(list)
    for r in rows:
        dur[r["n"]].append((r["e"] - r["s"]) / 1e3)
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print("%-32s n=%5d mean %7.2f us" % (n, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
