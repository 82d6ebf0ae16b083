#!/bin/bash
# round 6: the dataflow-across-launches experiment (GC_STREAM_DATAFLOW=1): parity of the streaming tests under it, then the bench programs
# with and without
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/${1:-r06df}; mkdir -p $OUT
GC_STREAM_DATAFLOW=1 timeout 1500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_stream_fuse.py tests/test_gpu_c_host.py tests/test_gpu_default_planner.py "tests/test_gpu_fuzz.py::test_queued_programs_across_the_scheduling_classes" -x -q -m gpu > $OUT/tests_df.log 2>&1
tail -n 6 $OUT/tests_df.log
for mode in 0 1; do
  GC_STREAM_DATAFLOW=$mode timeout 900 python scripts/bench_stream.py ${2:-ssa23:64 mixed:64 ed25519like:1024 uniform512:64 uniform4096:64 ed25519like:1} 2> $OUT/bench_df$mode.err | python scripts/sumjson.py | sed "s/^/DF=$mode /" | tee $OUT/bench_df$mode.txt
  tail -n 3 $OUT/bench_df$mode.err | cut -c1-300
done
