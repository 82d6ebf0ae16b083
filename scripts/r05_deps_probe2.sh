#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
for nm in ssa23:1024 ssa23:256 mixed:1024; do
  timeout 300 python scripts/bench_stream.py $nm 2>$OUT/deps_probe.err | python scripts/sumjson.py
  GC_STREAM_NO_DEPS=1 timeout 300 python scripts/bench_stream.py $nm 2>>$OUT/deps_probe.err | python scripts/sumjson.py | sed 's/^/NO_DEPS /'
done
for nm in ssa23:1024:native; do
  timeout 300 python scripts/bench_stream.py $nm 2>>$OUT/deps_probe.err | python scripts/sumjson.py
  GC_STREAM_NO_DEPS=1 timeout 300 python scripts/bench_stream.py $nm 2>>$OUT/deps_probe.err | python scripts/sumjson.py | sed 's/^/NO_DEPS /'
done
tail -n 5 $OUT/deps_probe.err
