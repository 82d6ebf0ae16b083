"""Reads the `[gc timeline]` lines of a GC_TRACE=3 run (stdin) and says, per garbling pass (a gc_stream), who waited for whom:
per group the GPU's idle time in front of its launch sequence, the sequence's duration (head -> kernel end -> bytes), how long
before the caller's wait the group was launched and how long the caller waited — group_timeline.py < stderr.log"""
import re, sys

passes, cur = [], []
last_no = -1
for l in sys.stdin:
    m = re.search(r"group (\d+) steps (\d+) host launch ([\d.-]+) ([\d.-]+) wait ([\d.-]+) ([\d.-]+) gpu head ([\d.-]+) kernel_end ([\d.-]+) bytes ([\d.-]+)", l)
    if not m:
        continue
    v = [float(x) for x in m.groups()]
    if cur and v[2] < cur[-1][2]:  # the host clock restarts with every stream
        passes.append(cur)
        cur = []
    cur.append(v)
if cur:
    passes.append(cur)
for pi, g in enumerate(passes):
    if len(g) < 8:
        continue
    span = max(x[8] for x in g) - g[0][6]
    kern = sum(x[7] - x[6] for x in g)
    idle = sum(max(0.0, g[i][6] - g[i - 1][7]) for i in range(1, len(g)))
    waits = [x[5] - x[4] for x in g if x[4] >= 0]
    lead = [x[4] - x[3] for x in g if x[4] >= 0]  # host: end of the launch sequence -> start of the wait for this group
    late = [x[8] - x[5] for x in g if x[4] >= 0]   # GPU bytes-there minus host wait end (~0 when the host really waited)
    copy = [x[8] - x[7] for x in g]
    print("pass %d: %d groups, %.0f steps/group, GPU span %.1f ms, head->kernel end %.1f ms in all (%.0f us avg), ctx stream idle between groups %.1f ms"
          % (pi, len(g), sum(x[1] for x in g) / len(g), span / 1e3, kern / 1e3, kern / len(g), idle / 1e3))
    print("   kernel end -> bytes there: avg %.0f us; host launch sequence avg %.0f us; host waited %.1f ms in all (avg %.0f us, %d of %d groups > 20 us)"
          % (sum(copy) / len(copy), sum(x[3] - x[2] for x in g) / len(g), sum(waits) / 1e3, sum(waits) / max(len(waits), 1),
             sum(w > 20 for w in waits), len(waits)))
    lead.sort()
    print("   launched before the caller's wait: median %.0f us, 10 %% below %.0f us; launched AT the wait (lead < 30 us): %d groups"
          % (lead[len(lead) // 2], lead[len(lead) // 10], sum(x < 30 for x in lead)))
