#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace of one streaming program + per-queue busy time (scripts/lanes.py)
# usage: scripts/profile_lanes.sh <tag> <program[:window[:native]]>
set -u
TAG=${1:-lanes}; P=${2:-ssa23}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rm -rf /tmp/kt_lanes
rocprofv3 --kernel-trace -f csv -d /tmp/kt_lanes -o kt -- python $REPO/scripts/bench_stream.py $P > $OUT/bench_lanes.log 2>&1
tail -1 $OUT/bench_lanes.log | cut -c1-400
python $REPO/scripts/lanes.py /tmp/kt_lanes garble | tee $OUT/lanes_garble.txt
python $REPO/scripts/lanes.py /tmp/kt_lanes eval | tee $OUT/lanes_eval.txt
