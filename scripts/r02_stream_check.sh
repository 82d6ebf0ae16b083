#!/bin/bash
# streaming parity tests + benchmark + per-phase trace of the evaluator (developer aid)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gpu_stream.py tests/test_host_mirror.py -x -q -m gpu 2>&1 | tail -2
python scripts/bench_stream.py 20000000 2>&1 | tail -1 | cut -c1-800
GC_TRACE=1 python scripts/bench_stream.py 2000000 2>&1 | grep "eval:" | tail -8
