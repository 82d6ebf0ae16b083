#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
for ds in 300 200 120 60; do
for l in 3 7; do
  GC_STREAM_DEEP_STEPS=$ds GC_STREAM_DEEP_LANES=$l timeout 300 python scripts/bench_stream.py ssa23:64 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/DEEP_STEPS $ds LANES $l /" | cut -c1-150
done
done
GC_STREAM_DEEP_STEPS=120 GC_STREAM_DEEP_LANES=7 timeout 300 python scripts/bench_stream.py ssa23:1024 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/DEEP_STEPS 120 LANES 7 /" | cut -c1-150
GC_STREAM_DEEP_STEPS=120 GC_STREAM_DEEP_LANES=7 GC_STREAM_NO_FOLLOW=1 timeout 300 python scripts/bench_stream.py ssa23:64 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/DEEP_STEPS 120 LANES 7 NO_FOLLOW /" | cut -c1-150
GC_STREAM_DEEP_LANES=7 GC_STREAM_NO_FOLLOW=1 timeout 300 python scripts/bench_stream.py ssa23:64 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/LANES 7 NO_FOLLOW /" | cut -c1-150
GC_STREAM_DEEP_STEPS=120 GC_STREAM_DEEP_LANES=7 timeout 300 python scripts/bench_stream.py mixed:64 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/DEEP_STEPS 120 LANES 7 /" | cut -c1-150
