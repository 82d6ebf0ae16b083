#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
for d in 80 128 200 320 640; do
  for nm in ssa23:64 ed25519like:1024; do
    GC_STREAM_FUSE_DEPTH=$d timeout 300 python scripts/bench_stream.py $nm 2>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/DEPTH $d /"
    GC_STREAM_FUSE_DEPTH=$d GC_STREAM_NO_DEPS=1 timeout 300 python scripts/bench_stream.py $nm 2>>$OUT/deps_probe.err | python scripts/sumjson.py | sed "s/^/DEPTH $d NO_DEPS /"
  done
done
