#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
for t in 3 5 8; do
  GC_STREAM_COPY_THREADS=$t timeout 300 python scripts/bench_stream.py ed25519like:1024 2>$OUT/probe7.err | python scripts/sumjson.py | sed "s/^/COPY_THREADS $t /" | cut -c1-170
  GC_STREAM_COPY_THREADS=$t timeout 300 python scripts/bench_stream.py ed25519like:1024:native 2>>$OUT/probe7.err | python scripts/sumjson.py | sed "s/^/COPY_THREADS $t /"
done
nproc; lscpu | grep -i "model name\|numa\|socket" | head -5
