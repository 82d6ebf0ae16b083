#!/bin/bash
# SQ counters of the fused kernels (instruction mix, LDS bank conflicts, VALU activity): supports the "issue-bound AES
# core" reading of DESIGN.md §4.  Separate --pmc passes, kernel-trace only (no sys/hip tracing).  Run through gpurun.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_sq
mkdir -p $OUT
cd /tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-iknp --no-graph --no-stream --no-config3 --no-host-api --no-synthetic --no-extra-rows"
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -f csv -d $OUT/$tag -o pmc -- python $REPO/bench.py $ARGS > $OUT/$tag.log 2>&1
done
python - <<PY > $OUT/summary.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "flat" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        print("   %-28s per dispatch %16.0f   (%d dispatches)" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
PY
rm -rf $OUT/*/*/*.db 2>/dev/null
cat $OUT/summary.txt
