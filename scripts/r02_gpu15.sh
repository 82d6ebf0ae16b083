#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_15; mkdir -p $OUT
python -m pytest tests/test_gpu_stream.py tests/test_host_mirror.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -6 $OUT/pytest.log
python scripts/bench_stream.py 20000000 2>&1 | tail -1
GC_TRACE=1 python scripts/bench_stream.py 2000000 2>&1 | grep "gc trace" | tail -8
