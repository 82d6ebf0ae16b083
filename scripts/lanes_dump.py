"""chronological list of the pass kernels of a streaming run (rocprofv3 --kernel-trace csv directory), second (timed) garble pass:
start us, duration us, queue, workgroups, kernel — lanes_dump.py DIR [garble|eval] > file"""
import csv, glob, sys

side = sys.argv[2] if len(sys.argv) > 2 else "garble"
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if ("flat_jobs" in n or "_coop<" in n or "k_garble_flat<" in n or "k_eval_flat<" in n) and side in n:
            wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), wg, n.split("(")[0][-48:]))
ev.sort()
if side == "garble":
    gaps = sorted(((ev[i + 1][0] - max(e[1] for e in ev[:i + 1][-8:]), i) for i in range(len(ev) // 4, 3 * len(ev) // 4)), reverse=True)
    ev = ev[gaps[0][1] + 1:]
t0 = ev[0][0]
for a, b, q, wg, n in ev:
    print("%.1f,%.1f,%s,%d,%s" % ((a - t0) / 1e3, (b - a) / 1e3, q, wg, n))
