#!/bin/bash
# round 6: the dataflow experiments side by side (GC_STREAM_DATAFLOW = off / 1 versions / 3 + ring of published units / 4 + persistent
# workgroups), default and eager launch policy (GC_STREAM_OPEN_GROUPS=1), Python host; every stream's SHA-256 is checked inside.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/${1:-r06dfm}; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
PROGS="ssa23:64 ssa23:256 mixed:64 ed25519like:1024 uniform512:64 uniform4096:64 ed25519like:1"
for cfg in "0 -" "1 -" "3 -" "4 -" "4 1" "0 1"; do
  set -- $cfg
  if [ "$2" = "-" ]; then unset GC_STREAM_OPEN_GROUPS; else export GC_STREAM_OPEN_GROUPS=$2; fi
  GC_STREAM_DATAFLOW=$1 timeout 900 python scripts/bench_stream.py $PROGS 2>>$OUT/err.txt | python scripts/sumjson.py | cut -c1-70 | sed "s/^/DATAFLOW=$1 OPEN_GROUPS=$2 /"
done | tee $OUT/matrix.txt
unset GC_STREAM_OPEN_GROUPS
for cfg in "0 -" "4 -" "4 1"; do
  set -- $cfg
  if [ "$2" = "-" ]; then unset GC_STREAM_OPEN_GROUPS; else export GC_STREAM_OPEN_GROUPS=$2; fi
  for p in ssa23:64:native ssa23:256:native ed25519like:1024:native; do
    GC_STREAM_DATAFLOW=$1 timeout 600 python scripts/bench_stream.py $p 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['program'], 'win', r['window'], 'C host: garble %.3g view %.3g async %.3g eval_blocks_pinned %.3g sha_ok %s' % (r['garble_gates_per_s'], r['garble_view_gates_per_s'], r['garble_async_gates_per_s'], r['eval_blocks_pinned_gates_per_s'], r['sha256_ok']))
" | sed "s/^/DATAFLOW=$1 OPEN_GROUPS=$2 /"
  done
done | tee $OUT/matrix_native.txt
tail -n 3 $OUT/err.txt | cut -c1-200
