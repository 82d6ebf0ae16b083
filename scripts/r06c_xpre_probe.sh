#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_garble_eval.py tests/test_gpu_go_transcript.py tests/test_gpu_config3.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
python scripts/r06c_tf_rows.py 2>&1 | tee gpurun_out/r06c_xpre_rows.txt
python scripts/bench_config3.py
