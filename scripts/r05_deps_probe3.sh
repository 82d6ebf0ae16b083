#!/bin/bash
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
for mode in deps nodeps; do
  cd /tmp; rm -rf /tmp/kt_lanes
  if [ $mode = nodeps ]; then export GC_STREAM_NO_DEPS=1; fi
  timeout 600 rocprofv3 --kernel-trace -f csv -d /tmp/kt_lanes -o kt -- python $REPO/scripts/bench_stream.py ssa23:1024 > $OUT/lanes_$mode.log 2>&1
  tail -1 $OUT/lanes_$mode.log | python $REPO/scripts/sumjson.py
  python $REPO/scripts/lanes.py /tmp/kt_lanes garble
  python $REPO/scripts/lanes_dump.py /tmp/kt_lanes garble > $OUT/lanes_dump_$mode.csv
done
