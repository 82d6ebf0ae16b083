#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_17; mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; echo "all gpu tests rc=$?"; tail -3 $OUT/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python scripts/bench_stream.py 130000000 > $OUT/stream_1e8.json 2> $OUT/stream.err; echo "stream rc=$?"; tail -1 $OUT/stream_1e8.json | cut -c1-900
timeout 900 python tests/ext_fuzz.py 400 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; grep -E "done|FAIL" $OUT/fuzz.log | head
