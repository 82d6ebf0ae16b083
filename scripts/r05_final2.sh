#!/bin/bash
# Run on the GPU box (through gpurun), on the tree after the second half of round 5 (serialiser through LDS, one-stream lone steps,
# six open groups, launch-ahead at every group boundary): the whole GPU suite, the full bench line, per-queue busy time and host
# stage cycles of the Ed25519-shaped program, the window-1 timeline, replicas on one GPU, a differential and a hostile-bytes
# fuzz run.  (The headline kernels did not change: profiles/r05_flat_* and latest_pmc.json of scripts/r05_final.sh stand.)
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05b
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r05b_gpu_suite.log 2>&1; echo "gpu suite rc=$?" >> $OUT/r05b_gpu_suite.log
tail -n 3 $OUT/r05b_gpu_suite.log
timeout 900 python bench.py > $OUT/r05b_bench_b1024.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05b/r05b_bench_b1024.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.4f  frac %.3f frac_read %.3f traffic %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["frac_read"], j["roofline"]["traffic"]))
s = j["stream"]
for k in ("ed25519like", "ssa23", "mixed", "uniform512", "uniform4096"):
    print(k, "%.3g %.3g %s" % (s[k]["garble_gates_per_s"], s[k]["eval_gates_per_s"], s[k].get("eval_blocks_gates_per_s")), s[k]["sha256_ok"])
print("big130 %.3g %.3g" % (s["steady_gates_per_s"], s["eval_steady_gates_per_s"]))
print("window1 %.3g" % s["ed25519like_window1"]["garble_gates_per_s"], "view %.3g" % s["ed25519like"]["garble_view_gates_per_s"])
print({k: {a: ("%.3g" % b if isinstance(b, float) else b) for a, b in v.items()} for k, v in s["native_host"].items()})
PY
bash scripts/profile_lanes.sh r05bl ed25519like:1024 > $OUT/r05b_stream_ed25519like_lanes.txt 2>&1
tail -n 5 $OUT/r05b_stream_ed25519like_lanes.txt
( GC_TRACE=1 timeout 600 python scripts/bench_stream.py ed25519like:1024:native 2>&1 >/dev/null | grep "host cycles" ) > $OUT/r05b_stream_host_stage_cycles.txt 2>&1
cat $OUT/r05b_stream_host_stage_cycles.txt | cut -c1-250
bash scripts/r05_w1_probe.sh > $OUT/w1.log 2>&1; cp gpurun_out/w1/w1_timeline.txt $OUT/r05b_w1_timeline.txt; tail -n 6 $OUT/r05b_w1_timeline.txt
timeout 300 python scripts/bench_stream_multi.py ed25519like 1 2 4 > $OUT/r05b_stream_replicas_ed25519like.jsonl 2> $OUT/replicas.err
cut -c1-330 $OUT/r05b_stream_replicas_ed25519like.jsonl
timeout 600 python tests/ext_fuzz.py 200 > $OUT/r05b_fuzz.log 2>&1; echo "fuzz rc=$?" >> $OUT/r05b_fuzz.log
tail -n 3 $OUT/r05b_fuzz.log | cut -c1-200
timeout 600 python tests/hostile_fuzz.py 600 > $OUT/r05b_hostile_fuzz.log 2>&1; echo "hostile fuzz rc=$?" >> $OUT/r05b_hostile_fuzz.log
tail -n 3 $OUT/r05b_hostile_fuzz.log | cut -c1-300
