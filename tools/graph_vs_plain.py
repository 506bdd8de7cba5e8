#!/usr/bin/env python3
"""Why a hipGraph replay of one frame takes longer than the same kernels launched one by one: from a rocprofv3 --kernel-trace CSV of
tools/frame100_bench.py (whose last 2000 steps are graph replays and the 2000 before them plain launches of the same step), the mean
duration of each kernel of the step and the mean gap in front of it, in both forms.   usage: graph_vs_plain.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lpf_" in r["Kernel_Name"]]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# kernels per step: the period of the name sequence at the end of the trace
names = [r["Kernel_Name"].split("(")[0].replace("void ", "") for r in rows]
k = next(p for p in range(1, 12) if names[-p * 8:] == (names[-p:] * 8))
graph, plain = rows[-steps * k:], rows[-(2 * steps + 20) * k - k:-(steps + 20) * k - k]          # (20 warm-up replays + the captured step between them)


def table(part, what):
    dur, gap = defaultdict(list), defaultdict(list)
    for i in range(k, len(part)):
        n = part[i]["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        dur[(i % k, n)].append(part[i]["e"] - part[i]["s"])
        gap[(i % k, n)].append(part[i]["s"] - part[i - 1]["e"])
    tot = (part[-1]["e"] - part[k]["s"]) / (len(part) // k - 1) / 1e3
    print("%s: %.2f us per step" % (what, tot))
    for key in sorted(dur):
        print("   %-46s dur %6.2f us   gap in front %6.2f us" % (key[1], sum(dur[key]) / len(dur[key]) / 1e3, sum(gap[key]) / len(gap[key]) / 1e3))


print("%d kernels per step" % k)
table(plain, "plain launches")
table(graph, "graph replays ")
