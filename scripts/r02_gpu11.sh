#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_11; mkdir -p $OUT
python -m pytest tests/test_gpu_garble_eval.py tests/test_gpu_fuzz.py tests/test_gpu_stream.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest.log
python bench.py --sweep > $OUT/sweep.json 2> $OUT/sweep.err; echo "sweep rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r02_11/sweep.json').read().strip().splitlines()[-1])
for r in d['sweep']:
    print("%-32s lds=%d g=%.3f e=%.3f AND/s=%.3g hbm=%.3f lds_arr=%.3f ok=%s"%(r['circuit'],r['wires_in_lds'],r['garble_ms'],r['eval_ms'],r['and_gates_per_s'],r['hbm_roofline_frac'],r['lds_array_frac'],r['outputs_ok']))
P
python scripts/bench_stream.py 20000000 2>&1 | tail -2
