#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_7; mkdir -p $OUT
python -m pytest tests/test_gpu_ot.py tests/test_gpu_egress.py tests/test_gpu_config3.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest.log
python scripts/bench_ot.py > $OUT/ot.json 2> $OUT/ot.err; echo "ot rc=$?"; cat $OUT/ot.json; tail -3 $OUT/ot.err
GC_COT_CLASSIC=1 python scripts/bench_ot.py > $OUT/ot_classic.json 2> $OUT/ot_classic.err; echo "ot classic rc=$?"; python -c "import json;d=json.load(open('$OUT/ot_classic.json'));print(d['cot'])"
