#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace summary of the streaming programs of scripts/bench_stream.py
# (settle "launch-bound vs latency-bound" for the streaming kernels with a profile).
# usage: scripts/profile_stream.sh <tag> [programs...]
set -u
TAG=${1:-stream}; shift || true
PROGS=${*:-uniform512 uniform4096 mixed big}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
for P in $PROGS; do
  N=${P%%:*}
  rocprofv3 --kernel-trace --stats -f csv -d $OUT/kt_$N -o kt -- python $REPO/scripts/bench_stream.py $P > $OUT/bench_$N.log 2>&1
  find $OUT/kt_$N -name "*kernel_stats.csv" -exec cp {} $OUT/${N}_kernel_stats.csv \;
  rm -rf $OUT/kt_$N
  tail -1 $OUT/bench_$N.log | cut -c1-600
  head -12 $OUT/${N}_kernel_stats.csv | cut -c1-220
done
