#!/bin/bash
# host-side knobs of the default streaming engine once more on the shipped tree: deep lanes (3 = default) and the deep-step threshold
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
PROGS="ssa23:64 mixed:64 ed25519like:1024"
for lanes in 3 5 7; do
  for deep in 300 150; do
    GC_STREAM_DEEP_LANES=$lanes GC_STREAM_DEEP_STEPS=$deep timeout 600 python scripts/bench_stream.py $PROGS 2>/dev/null | python scripts/sumjson.py | cut -c1-90 | sed "s/^/lanes=$lanes deep_steps=$deep /"
  done
done | tee gpurun_out/r06c_lanes_sweep.txt
