#!/bin/bash
# Run on the GPU box (through gpurun): the host's stage cycles of the C driver on the Ed25519-shaped program (window 1 024) and
# the aggregate of 1 / 2 / 4 C-driver processes on ONE GPU (replicas: config 5 does not shard).
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/host
mkdir -p $OUT
cd $REPO
GC_TRACE=1 python scripts/bench_stream.py ed25519like:1024:native > $OUT/native.json 2> $OUT/native.err
grep "host cycles" $OUT/native.err
cut -c1-900 $OUT/native.json
python scripts/bench_stream_multi.py ed25519like 1 2 4 > $OUT/replicas.jsonl 2> $OUT/replicas.err
cat $OUT/replicas.jsonl | cut -c1-600
tail -3 $OUT/replicas.err
