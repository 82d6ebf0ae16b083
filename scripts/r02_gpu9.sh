#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r02_9; mkdir -p $OUT
python bench.py --sweep > $OUT/sweep.json 2> $OUT/sweep.err; echo "sweep rc=$?"; tail -2 $OUT/sweep.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r02_9/sweep.json').read().strip().splitlines()[-1])
print(d['sweep_wall_s'])
for r in d['sweep']:
    print("%-32s lds=%d live=%s g=%.3f e=%.3f AND/s=%.3g hbm=%.3f read=%.3f lds_arr=%.3f ok=%s"%(r['circuit'],r['wires_in_lds'],r['live_labels'],r['garble_ms'],r['eval_ms'],r['and_gates_per_s'],r['hbm_roofline_frac'],r['hbm_read_roofline_frac'],r['lds_array_frac'],r['outputs_ok']))
P
