#!/bin/bash
# Run on the GPU box (through gpurun): where the UNCHANGED caller's 184 us per instruction go (window 1: begin + finish per
# step) — the host's stage cycles (GC_TRACE) and a kernel timeline of the run.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/w1
mkdir -p $OUT
cd $REPO
python scripts/bench_stream.py ed25519like:1 > $OUT/plain.json 2> $OUT/plain.err
cut -c1-400 $OUT/plain.json
GC_TRACE=2 python scripts/bench_stream.py ed25519like:1 2> $OUT/trace.err > $OUT/trace.json
grep "host cycles" $OUT/trace.err | tail -4
python scripts/agg_trace.py < $OUT/trace.err | head -12
cd /tmp
rocprofv3 --kernel-trace -f csv -d $OUT/kt -o kt -- python $REPO/scripts/bench_stream.py ed25519like:1 > $OUT/kt.log 2>&1
python $REPO/scripts/w1_timeline.py $OUT/kt | tee $OUT/w1_timeline.txt
rm -rf $OUT/kt
