#!/bin/bash
# HBM-wire kernels (k_garble_fused / k_eval_fused) on the synthetic levelised circuits of SURVEY §8d, W = 1024, f = 0 and
# f = 0.17: rocprofv3 kernel trace + PMC (HBM bytes: FETCH_SIZE / WRITE_SIZE in separate passes; LDS and VALU activity).
# the "4.5 TB/s, 56 % of HBM" figure of the f = 0 row backed by counters.  Run through gpurun.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_synth
mkdir -p $OUT
cd /tmp
CMD="python $REPO/scripts/sweep_synthetic.py 1024 131072 1024:0,1024:0.17"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -f csv -d $OUT/$tag -o pmc -- $CMD > $OUT/$tag.log 2>&1
done
python - <<PY > $OUT/summary.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void gc::", "")[:48]
        if "fused" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("per dispatch, averaged over the dispatches of BOTH circuits (f = 0 and f = 0.17; 4 passes each); FETCH_SIZE / WRITE_SIZE in KiB,")
print("FETCH_SIZE under-reports wide coalesced reads 2x on gfx950 (MI355X_MICROARCH.md): HBM bytes = 2 x FETCH + WRITE")
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print("   %-28s per dispatch %18.0f   (%d dispatches)" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
PY
# the same split per circuit: dispatches come in order f = 0 (4 garble + 4 eval) then f = 0.17
python - <<PY >> $OUT/summary.txt
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void gc::", "")[:48]
        if "fused" not in k or r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"): continue
        rows[(k, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
print()
print("HBM counters per circuit (first half of the dispatches: f = 0; second half: f = 0.17), KiB per dispatch")
for (k, c), v in sorted(rows.items()):
    v.sort()
    h = len(v) // 2
    a = sum(x for _, x in v[:h]) / max(h, 1); b = sum(x for _, x in v[h:]) / max(len(v) - h, 1)
    print("   %-48s %-11s f=0: %14.0f   f=0.17: %14.0f" % (k, c, a, b))
PY
rm -rf $OUT/*/*/*.db $OUT/kt 2>/dev/null
cat $OUT/summary.txt; head -8 $OUT/kernel_stats.csv | cut -c1-200
