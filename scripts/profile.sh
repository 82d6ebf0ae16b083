#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace profile of the bench command, then the two PMC
# passes for HBM bytes (separate runs, as MI355X_MICROARCH.md §HBM prescribes).
# usage: scripts/profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
ARGS="--steps 100 --warmup 20 --no-cpu-baseline --no-iknp --no-stream --no-config3 --no-host-api --no-synthetic --no-extra-rows $*"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt -o kt -- python $REPO/bench.py $ARGS > $OUT/bench_kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/kt -name "*kernel_trace.csv" -exec sh -c 'head -400 "$1" > '$OUT'/kernel_trace_head.csv' _ {} \;
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_r -o pmc -- python $REPO/bench.py $ARGS > $OUT/bench_pmc_r.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/pmc_w -o pmc -- python $REPO/bench.py $ARGS > $OUT/bench_pmc_w.log 2>&1
# batch / schedule / key bytes of this run (defaults of bench.py unless overridden in the arguments)
B=1024; S=1; K=32
set -- $ARGS
while [ $# -gt 0 ]; do case "$1" in --batch) B=$2; shift;; --schedule) S=$2; shift;; --key-bytes) K=$2; shift;; esac; shift; done
python $REPO/scripts/pmc_summary.py $OUT $B $S $K $TAG > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/kt/*/*.db $OUT/pmc_r/*/*.db $OUT/pmc_w/*/*.db 2>/dev/null
du -sh $OUT; ls $OUT
