"""Where a window-1 streamed step spends its time on the GPU (rocprofv3 --kernel-trace csv directory of
`bench_stream.py ed25519like:1`): per kernel name count / average duration, and the gaps between the garbling kernel of a step,
its serialiser and the garbling kernel of the next step — w1_timeline.py DIR"""
import csv, glob, sys, collections

ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]))
ev.sort()
dur = collections.defaultdict(float)
cnt = collections.Counter()
for a, b, n in ev:
    dur[n] += b - a
    cnt[n] += 1
for n in sorted(dur, key=lambda n: -dur[n])[:12]:
    print("%-62s n=%7d avg %8.2f us total %9.2f ms" % (n, cnt[n], dur[n] / cnt[n] / 1e3, dur[n] / 1e6))
# the second half of the trace is the timed pass: garble kernel -> serialiser -> next garble kernel
half = ev[len(ev) // 2:]
g2s, s2g, gd, sd, per = [], [], [], [], []
last_g = last_s = None
for a, b, n in half:
    if "garble_flat_jobs" in n:
        if last_s is not None:
            s2g.append(a - last_s[1])
        if last_g is not None:
            per.append(a - last_g[0])
        last_g = (a, b)
        gd.append(b - a)
    elif "serialise" in n or "k_ser" in n:
        if last_g is not None:
            g2s.append(a - last_g[1])
        last_s = (a, b)
        sd.append(b - a)


def med(v):
    v = sorted(v)
    return v[len(v) // 2] / 1e3 if v else float("nan")


def avg(v):
    return sum(v) / len(v) / 1e3 if v else float("nan")


print("timed half: %d garbling kernels" % len(gd))
print("garble kernel        avg %7.2f us  median %7.2f" % (avg(gd), med(gd)))
print("kernel -> serialiser avg %7.2f us  median %7.2f" % (avg(g2s), med(g2s)))
print("serialiser           avg %7.2f us  median %7.2f" % (avg(sd), med(sd)))
print("serialiser -> next   avg %7.2f us  median %7.2f   (D2H copy, host wake-up, next begin, H2D copy, launch)" % (avg(s2g), med(s2g)))
print("step period          avg %7.2f us  median %7.2f" % (avg(per), med(per)))
