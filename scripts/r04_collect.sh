#!/bin/bash
# Run on the GPU box (through gpurun): everything round 4 files under profiles/ — the queue-overlap micro-benchmark, kernel
# trace + PMC passes of the headline (scripts/profile.sh), kernel traces of the streaming programs, the lanes' busy time, and
# the full bench line (with the PMC traffic of THIS build: latest_pmc.json is refreshed first).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04
mkdir -p $OUT
cd $REPO
( cd tools && ./queue_overlap_ubench 2 4 0; echo "---- GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 ./queue_overlap_ubench 4 4 0 ) > $OUT/r04_queue_overlap_ubench.txt 2>&1
bash scripts/profile.sh r04 > $OUT/profile.log 2>&1
cp gpurun_out/prof_r04/kernel_stats.csv $OUT/r04_flat_kernel_stats.csv
cp gpurun_out/prof_r04/pmc_summary.txt $OUT/r04_flat_pmc_summary.txt
cp gpurun_out/prof_r04/latest_pmc.json $OUT/latest_pmc.json && cp $OUT/latest_pmc.json profiles/latest_pmc.json
bash scripts/profile_stream.sh r04s ssa23 ed25519like:1024 mixed > $OUT/profile_stream.log 2>&1
for p in ssa23 ed25519like mixed; do cp gpurun_out/prof_r04s/${p}_kernel_stats.csv $OUT/r04_stream_${p}_kernel_stats.csv; done
bash scripts/profile_lanes.sh r04l ssa23 > $OUT/r04_stream_ssa23_lanes.txt 2>&1
bash scripts/profile_lanes.sh r04m ed25519like:1024 > $OUT/r04_stream_ed25519like_lanes.txt 2>&1
python bench.py > $OUT/r04_bench_b1024.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04/r04_bench_b1024.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.4f  frac %.3f frac_read %.3f traffic %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["frac_read"], j["roofline"]["traffic"]))
s = j["stream"]
for k in ("ed25519like", "ssa23", "mixed"):
    print(k, "%.3g %.3g" % (s[k]["garble_gates_per_s"], s[k]["eval_gates_per_s"]), s[k]["sha256_ok"])
PY
