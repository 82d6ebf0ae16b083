#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_8; mkdir -p $OUT
python -m pytest tests/test_gpu_ot.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.log
python scripts/bench_ot.py > $OUT/ot.json 2> $OUT/ot.err; echo "ot rc=$?"; python -c "import json;d=json.load(open('$OUT/ot.json'));print(d['kos'], d['cot'])"
bash scripts/profile_ot.sh r02_ot
