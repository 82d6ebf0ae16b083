#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for t in 8 4 2; do
  export GC_PLAN_PART_TERMS=$t
  echo "== GC_PLAN_PART_TERMS=$t"
  python -m pytest tests/test_gpu_garble_eval.py tests/test_gpu_go_transcript.py tests/test_gpu_config3.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
  python scripts/r06c_tf_rows.py 2>&1 | cut -c1-200
done | tee gpurun_out/r06c_parts_probe.txt
