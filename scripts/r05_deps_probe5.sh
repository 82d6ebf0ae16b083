#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
for q in 8 16; do
for l in 3 5 7; do
  GPU_MAX_HW_QUEUES=$q GC_STREAM_DEEP_LANES=$l GC_STREAM_NO_DEPS=1 timeout 300 python scripts/bench_stream.py ssa23:64 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/HWQ $q LANES $l NO_DEPS /" | cut -c1-150
  GPU_MAX_HW_QUEUES=$q GC_STREAM_DEEP_LANES=$l timeout 300 python scripts/bench_stream.py ssa23:64 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/HWQ $q LANES $l /" | cut -c1-150
done
done
GPU_MAX_HW_QUEUES=16 GC_STREAM_DEEP_LANES=7 GC_STREAM_NO_DEPS=1 timeout 300 python scripts/bench_stream.py ssa23:1024 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/HWQ 16 LANES 7 NO_DEPS /" | cut -c1-150
GPU_MAX_HW_QUEUES=16 GC_STREAM_DEEP_LANES=7 timeout 300 python scripts/bench_stream.py ssa23:1024 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/HWQ 16 LANES 7 /" | cut -c1-150
GPU_MAX_HW_QUEUES=16 GC_STREAM_DEEP_LANES=7 GC_TRACE=1 timeout 300 python scripts/bench_stream.py ssa23:64 2>&1 >/dev/null | grep "deep lane" | sort | uniq -c
