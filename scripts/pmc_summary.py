#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs per kernel (sum over dispatches).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md §HBM) — both raw and corrected figures are printed."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
per = {}  # (pass, kernel prefix) -> KiB per dispatch
for tag, counter in (("pmc_r", "FETCH_SIZE"), ("pmc_w", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    tot = defaultdict(float)
    cnt = defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:80]
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    print("== %s (KiB, summed over dispatches) ==" % counter)
    for k in sorted(tot, key=lambda x: -tot[x])[:12]:
        corr = 2.0 if counter == "FETCH_SIZE" else 1.0
        print("%-80s dispatches=%6d  raw=%14.0f KiB  per-dispatch=%10.1f KiB  corrected(x%g)=%10.1f KiB" % (
            k, cnt[k], tot[k], tot[k] / cnt[k], corr, corr * tot[k] / cnt[k]))
    for k in tot:
        per[(tag, k)] = tot[k] / cnt[k]

# machine-readable per-launch HBM traffic of the dominant kernels (bench.py reads profiles/latest_pmc.json)
if len(sys.argv) > 5:
    import json
    batch, schedule, key_bytes, tag_name = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]

    def pick(tag, frag):
        for (t, k), v in per.items():
            if t == tag and frag in k:
                return v
        return None

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_build_hash  # the counters describe THIS build of the kernels: bench.py drops them otherwise
    j = {"batch": batch, "schedule": schedule, "key_bytes": key_bytes, "kernel_build_hash": kernel_build_hash(),
         "source": "profiles/%s_flat_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                   "FETCH_SIZE x2 per MI355X_MICROARCH.md)" % tag_name}
    for nm in ("garble", "eval"):
        r, w = pick("pmc_r", "k_%s_" % nm), pick("pmc_w", "k_%s_" % nm)
        if r is not None and w is not None:
            j["%s_fetch_kib_raw" % nm] = round(r, 1)
            j["%s_write_kib" % nm] = round(w, 1)
            j["%s_hbm_bytes_per_launch" % nm] = int((2.0 * r + w) * 1024)
    json.dump(j, open(os.path.join(out, "latest_pmc.json"), "w"), indent=1)
