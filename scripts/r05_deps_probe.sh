#!/bin/bash
# units that wait inside a launch (GC_STREAM_NO_DEPS=1: off): parity tests, then the programs with and without
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_stream_fuse.py -q 2>&1 | tail -n 15
timeout 900 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -n 5
for nm in ssa23:64 mixed:64 ed25519like:1024 uniform512:64; do
  timeout 300 python scripts/bench_stream.py $nm 2>$OUT/deps_probe.err | python scripts/sumjson.py
  GC_STREAM_NO_DEPS=1 timeout 300 python scripts/bench_stream.py $nm 2>>$OUT/deps_probe.err | python scripts/sumjson.py | sed 's/^/NO_DEPS /'
done
GC_STREAM_DEP_FENCES=1 timeout 300 python scripts/bench_stream.py ssa23:64 2>>$OUT/deps_probe.err | python scripts/sumjson.py | sed 's/^/FENCES /'
for nm in ssa23:64:native ed25519like:1024:native; do
  timeout 300 python scripts/bench_stream.py $nm 2>>$OUT/deps_probe.err | python scripts/sumjson.py
done
tail -n 5 $OUT/deps_probe.err
