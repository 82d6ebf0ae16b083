#!/bin/bash
# Run on the GPU box (through gpurun), on the final tree of round 5: PMC passes of the headline for THIS build of the kernels
# (scripts/profile.sh -> profiles/latest_pmc.json), the whole GPU suite, the full bench line, the hostile-bytes campaign's
# pinned device-matched mode (the one the 3600 s limit cut off in scripts/r05_collect2.sh).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
bash scripts/profile.sh r05f > $OUT/profile.log 2>&1
cp gpurun_out/prof_r05f/kernel_stats.csv $OUT/r05_flat_kernel_stats.csv
cp gpurun_out/prof_r05f/pmc_summary.txt $OUT/r05_flat_pmc_summary.txt
cp gpurun_out/prof_r05f/latest_pmc.json $OUT/latest_pmc.json && cp $OUT/latest_pmc.json profiles/latest_pmc.json
tail -n 3 $OUT/r05_flat_pmc_summary.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?" >> $OUT/gpu_suite.log
tail -n 4 $OUT/gpu_suite.log
timeout 900 python bench.py > $OUT/r05_bench_b1024.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05/r05_bench_b1024.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.4f  frac %.3f frac_read %.3f traffic %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["frac_read"], j["roofline"]["traffic"]))
s = j["stream"]
for k in ("ed25519like", "ssa23", "mixed"):
    print(k, "%.3g %.3g %s" % (s[k]["garble_gates_per_s"], s[k]["eval_gates_per_s"], s[k].get("eval_blocks_gates_per_s")), s[k]["sha256_ok"])
PY
timeout 900 python - > $OUT/r05_hostile_fuzz_pinned.log 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from tests import hostile_fuzz as h
r = h.run_device(600, 5, log=print, pinned=True)
print("read buffers matched on the device (pinned): %d mutants, %s, kinds %s, device blocks / fallbacks %s — no violation" % r, flush=True)
PY
echo "rc=$?" >> $OUT/r05_hostile_fuzz_pinned.log
tail -n 2 $OUT/r05_hostile_fuzz_pinned.log | cut -c1-300
