#!/bin/bash
# Run on the GPU box (through gpurun) on the FINAL tree of round 6, as the last action that touches a hashed file: kernel stats + the
# two PMC passes of the headline (scripts/profile.sh -> profiles/latest_pmc.json, tied to device sources + plan fingerprint), SQ
# counters, the whole GPU suite, the full bench line, the C host at config 5's size ten times over under a parent process (BENCH_r05's
# failure), a differential fuzz run on the product's default planner and a hostile-bytes campaign that runs to its end.
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
bash scripts/profile.sh $TAG > $OUT/profile.log 2>&1
cp gpurun_out/prof_$TAG/kernel_stats.csv $OUT/${TAG}_flat_kernel_stats.csv
cp gpurun_out/prof_$TAG/pmc_summary.txt $OUT/${TAG}_flat_pmc_summary.txt
cp gpurun_out/prof_$TAG/latest_pmc.json $OUT/latest_pmc.json
tail -n 3 $OUT/${TAG}_flat_pmc_summary.txt; cat $OUT/latest_pmc.json
bash scripts/profile_sq.sh > $OUT/profile_sq.log 2>&1; cp gpurun_out/prof_sq/summary.txt $OUT/${TAG}_flat_sq_counters.txt; tail -n 4 $OUT/${TAG}_flat_sq_counters.txt
cp $OUT/latest_pmc.json profiles/latest_pmc.json   # (this box's copy: the bench run below reads it)
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_suite.log 2>&1; echo "gpu suite rc=$?" >> $OUT/${TAG}_gpu_suite.log
tail -n 3 $OUT/${TAG}_gpu_suite.log
# (the persistent-workgroup form of the dataflow experiment needs 16 hardware queues from the host: skipped in the suite above)
GPU_MAX_HW_QUEUES=16 timeout 900 python -m pytest tests/test_gpu_dataflow.py -m gpu -q > $OUT/${TAG}_gpu_dataflow_hwq16.log 2>&1; echo "rc=$?" >> $OUT/${TAG}_gpu_dataflow_hwq16.log
tail -n 2 $OUT/${TAG}_gpu_dataflow_hwq16.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_b1024.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err
python - "$OUT/${TAG}_bench_b1024.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("value %.4g  ms/step %.4f  frac %.3f frac_read %.3f traffic %s (%s) hbm_counter_frac %s" % (j["value"], j["ms_per_step"], r["frac"], r["frac_read"], r["traffic"], r["traffic_source"], r["hbm_counter_frac"]))
print("level_launch %.3g  n1_batch8192 %.3g  config3 %.3g  and_chain ns %.0f" % (j["level_launch"]["and_gates_per_s"], j["n1_batch8192"]["and_gates_per_s"], j["config3"]["and_gates_per_s"], j["and_chain_10000"]["ns_per_and_per_instance_garble"]))
s = j["stream"]
for k in ("ed25519like", "ssa23", "mixed", "uniform512", "uniform4096"):
    print(k, "%.3g %.3g %s" % (s[k]["garble_gates_per_s"], s[k]["eval_gates_per_s"], s[k].get("eval_blocks_gates_per_s")), s[k]["sha256_ok"])
print("big130 %.3g %.3g" % (s["steady_gates_per_s"], s["eval_steady_gates_per_s"]))
print("window1 %.3g" % s["ed25519like_window1"]["garble_gates_per_s"], "view %.3g" % s["ed25519like"]["garble_view_gates_per_s"])
print("dataflow_experiment", {k: v for k, v in s.get("dataflow_experiment", {}).items() if k != "note"})
print({k: {a: ("%.3g" % b if isinstance(b, float) else b) for a, b in v.items()} for k, v in s["native_host"].items()})
PY
timeout 900 python scripts/r06_native_loop.py 10 > $OUT/${TAG}_native_host_loop.jsonl 2> $OUT/native_loop.err; echo "rc=$?" >> $OUT/${TAG}_native_host_loop.jsonl
tail -n 4 $OUT/${TAG}_native_host_loop.jsonl | cut -c1-250
GC_FUZZ_DEFAULT_PLANNER=1 timeout 900 python tests/ext_fuzz.py 200 > $OUT/${TAG}_fuzz_default_planner.log 2>&1; echo "fuzz rc=$?" >> $OUT/${TAG}_fuzz_default_planner.log
tail -n 3 $OUT/${TAG}_fuzz_default_planner.log | cut -c1-200
timeout 2400 python tests/hostile_fuzz.py 300 > $OUT/${TAG}_hostile_fuzz.log 2>&1; echo "hostile fuzz rc=$?" >> $OUT/${TAG}_hostile_fuzz.log
tail -n 2 $OUT/${TAG}_hostile_fuzz.log | cut -c1-300
