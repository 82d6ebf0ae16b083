#!/bin/bash
# Run on the GPU box (through gpurun): everything round 5 files under profiles/ — kernel trace + PMC passes of the headline
# (scripts/profile.sh: latest_pmc.json for THIS build of the kernels), SQ counters, kernel stats and per-queue busy time of the
# streaming programs with chain fusion, the stage-cycle laps of both streaming hosts, the hostile-bytes fuzz incl. the
# device-matched read buffers, the differential fuzz, the fixed-cost probe, and the full bench line.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
bash scripts/profile.sh r05 > $OUT/profile.log 2>&1
cp gpurun_out/prof_r05/kernel_stats.csv $OUT/r05_flat_kernel_stats.csv
cp gpurun_out/prof_r05/pmc_summary.txt $OUT/r05_flat_pmc_summary.txt
cp gpurun_out/prof_r05/latest_pmc.json $OUT/latest_pmc.json && cp $OUT/latest_pmc.json profiles/latest_pmc.json
bash scripts/profile_sq.sh > $OUT/profile_sq.log 2>&1
cp gpurun_out/prof_sq/summary.txt $OUT/r05_flat_sq_counters.txt
bash scripts/profile_stream.sh r05s ssa23 ed25519like:1024 mixed > $OUT/profile_stream.log 2>&1
for p in ssa23 ed25519like mixed; do cp gpurun_out/prof_r05s/${p}_kernel_stats.csv $OUT/r05_stream_${p}_kernel_stats.csv; done
bash scripts/profile_lanes.sh r05l ssa23 > $OUT/r05_stream_ssa23_lanes.txt 2>&1
bash scripts/profile_lanes.sh r05m ed25519like:1024 > $OUT/r05_stream_ed25519like_lanes.txt 2>&1
( GC_TRACE=1 timeout 600 python scripts/bench_stream.py ed25519like:1024 2>&1 >/dev/null | grep "host cycles" ) > $OUT/r05_stream_host_stage_cycles.txt 2>&1
timeout 300 python scripts/probe_fixed_cost.py > $OUT/r05_fixed_cost_probe.json 2>&1
timeout 2400 python tests/hostile_fuzz.py 10000 > $OUT/r05_hostile_fuzz.log 2>&1; echo "hostile fuzz rc=$?" >> $OUT/r05_hostile_fuzz.log
timeout 1500 python tests/ext_fuzz.py 600 > $OUT/r05_fuzz.log 2>&1; echo "fuzz rc=$?" >> $OUT/r05_fuzz.log
python bench.py > $OUT/r05_bench_b1024.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05/r05_bench_b1024.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.4f  frac %.3f frac_read %.3f traffic %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["frac_read"], j["roofline"]["traffic"]))
print("level_launch %.4g" % j["level_launch"]["and_gates_per_s"])
s = j["stream"]
for k in ("ed25519like", "ssa23", "mixed"):
    print(k, "%.3g %.3g %s" % (s[k]["garble_gates_per_s"], s[k]["eval_gates_per_s"], s[k].get("eval_blocks_gates_per_s")), s[k]["sha256_ok"])
print("window1 %.3g" % s["ed25519like_window1"]["garble_gates_per_s"], "view %.3g" % s["ed25519like"]["garble_view_gates_per_s"])
print({k: {a: ("%.3g" % b if isinstance(b, float) else b) for a, b in v.items()} for k, v in s["native_host"].items() if k in ("ed25519like", "ssa23")})
PY
tail -3 $OUT/r05_hostile_fuzz.log $OUT/r05_fuzz.log
