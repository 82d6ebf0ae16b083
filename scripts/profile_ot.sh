#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace of the OT benchmark (4 Mi OTs: IKNP, bit-COT, COT pads,
# KOS check), then the two PMC passes for the HBM bytes of the same kernels (separate runs).
# usage: scripts/profile_ot.sh <tag>
set -u
TAG=${1:-ot}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python $REPO/scripts/bench_ot.py > $OUT/bench_kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_r -o pmc -- python $REPO/scripts/bench_ot.py > $OUT/bench_pmc_r.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_w -o pmc -- python $REPO/scripts/bench_ot.py > $OUT/bench_pmc_w.log 2>&1
python $REPO/scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/kt/*/*.db $OUT/pmc_r/*/*.db $OUT/pmc_w/*/*.db 2>/dev/null
head -12 $OUT/kernel_stats.csv; cat $OUT/pmc_summary.txt | head -30; tail -1 $OUT/bench_kt.log | cut -c1-900
