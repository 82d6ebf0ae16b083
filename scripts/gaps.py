"""idle time between the pass kernels of a streaming run (rocprofv3 --kernel-trace csv directory): gaps.py DIR"""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "flat_jobs" in n or "_coop<" in n:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "garble" if "garble" in n else "eval", "coop" if "coop" in n else "jobs"))
ev.sort()
for side in ("garble", "eval"):
    e = [x for x in ev if x[2] == side]
    e = e[len(e) // 2:]  # the second (timed) run
    busy = sum(b - a for a, b, _, _ in e)
    span = e[-1][1] - e[0][0]
    gaps = [(e[i + 1][0] - e[i][1]) / 1e3 for i in range(len(e) - 1)]
    after_coop = [g for i, g in enumerate(gaps) if e[i][3] == "coop"]
    before_coop = [g for i, g in enumerate(gaps) if e[i + 1][3] == "coop" and e[i][3] != "coop"]
    rest = [g for i, g in enumerate(gaps) if e[i][3] != "coop" and e[i + 1][3] != "coop"]
    f = lambda v: "n=%d avg %.1f us total %.1f ms" % (len(v), sum(v) / max(len(v), 1), sum(v) / 1e3)
    print(side, "kernels", len(e), "span %.1f ms busy %.1f ms" % (span / 1e6, busy / 1e6))
    print("   gaps after a big step:", f(after_coop)); print("   gaps before a big step:", f(before_coop)); print("   gaps between groups:", f(rest))
    big = sorted(gaps)[-8:]
    print("   largest gaps (us):", ["%.0f" % g for g in big])
