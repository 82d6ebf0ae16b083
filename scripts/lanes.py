"""per-queue busy time of the pass kernels of a streaming run (rocprofv3 --kernel-trace csv directory): which hardware queue
ran what, how long each was busy, and how much of the run the ctx stream's queue sat idle: lanes.py DIR [garble|eval]"""
import csv, glob, sys
from collections import defaultdict

side = sys.argv[2] if len(sys.argv) > 2 else "garble"
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if ("flat_jobs" in n or "_coop<" in n or "k_garble_flat<" in n or "k_eval_flat<" in n) and side in n:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n.split("(")[0][-40:]))
ev.sort()
if side == "garble":  # the second (timed) pass: the larger half by time
    mid = (ev[0][0] + ev[-1][1]) // 2
    # a gap between the two passes: split at the largest idle stretch near the middle
    gaps = sorted(((ev[i + 1][0] - max(e[1] for e in ev[:i + 1][-8:]), i) for i in range(len(ev) // 4, 3 * len(ev) // 4)), reverse=True)
    cut = gaps[0][1] + 1
    ev = ev[cut:]
span = ev[-1][1] - ev[0][0]
print(side, "pass kernels", len(ev), "span %.1f ms" % (span / 1e6))
byq = defaultdict(list)
for e in ev:
    byq[e[2]].append(e)
for q, es in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(b - a for a, b, _, _ in es)
    long_ = [b - a for a, b, _, _ in es if b - a > 400_000]
    gaps = [es[i + 1][0] - es[i][1] for i in range(len(es) - 1)]
    gaps_sorted = sorted(gaps)
    print("  queue %s: %d kernels, busy %.1f ms (%.0f %%), avg %.0f us, %d longer than 0.4 ms (%.1f ms); gaps: median %.1f us, total %.1f ms, > 100 us: %d (%.1f ms)" % (
        q, len(es), busy / 1e6, 100.0 * busy / span, busy / len(es) / 1e3, len(long_), sum(long_) / 1e6,
        gaps_sorted[len(gaps) // 2] / 1e3 if gaps else 0, sum(gaps) / 1e6, sum(1 for g in gaps if g > 100_000), sum(g for g in gaps if g > 100_000) / 1e6))
