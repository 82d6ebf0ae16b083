#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_18; mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; echo "all gpu tests rc=$?"; tail -3 $OUT/pytest_all.log
python scripts/bench_stream.py 130000000 > $OUT/stream_1e8.json 2> $OUT/stream.err; echo "stream rc=$?"; tail -1 $OUT/stream_1e8.json | cut -c1-900
python bench.py --sweep > $OUT/sweep.json 2>/dev/null; echo "sweep rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r02_18/sweep.json').read().strip().splitlines()[-1])
for r in d['sweep']:
    print("%-32s lds=%d g=%.3f e=%.3f AND/s=%.3g"%(r['circuit'],r['wires_in_lds'],r['garble_ms'],r['eval_ms'],r['and_gates_per_s']))
P
timeout 600 python tests/ext_fuzz.py 150 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; grep -E "done|FAIL" $OUT/fuzz.log | head
