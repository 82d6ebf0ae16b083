#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of the level-launch row: per kernel name the mean duration, and the mean gap between the end of
one level kernel and the start of the next on the same queue (the price of a dependent graph node)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = defaultdict(list)
gaps = defaultdict(list)
prev = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev and prev[0] == name and "level" in name:
        gaps[name].append(s - prev[1])
    prev = (name, e)
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:6]:
    g = gaps.get(n)
    print("%-60s n=%6d mean %8.2f us  p50 %8.2f us  gap to the next %s" % (
        n, len(d), sum(d) / len(d) / 1e3, sorted(d)[len(d) // 2] / 1e3,
        "%.2f us (p50 %.2f)" % (sum(g) / len(g) / 1e3, sorted(g)[len(g) // 2] / 1e3) if g else "-"))
