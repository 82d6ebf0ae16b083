#!/bin/bash
# second half of scripts/r05_collect.sh (the first run's bench line failed on a NameError, its hostile-bytes campaign ran out of
# time before the device-matched buffers): the full bench line of the final tree, the stage cycles, the hostile-bytes campaign
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
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
( GC_TRACE=1 timeout 600 python scripts/bench_stream.py ed25519like:1024 2>&1 >/dev/null | grep "host cycles" ) > $OUT/r05_stream_host_stage_cycles.txt 2>&1
timeout 4200 python tests/hostile_fuzz.py 6000 > $OUT/r05_hostile_fuzz.log 2>&1; echo "hostile fuzz rc=$?" >> $OUT/r05_hostile_fuzz.log
tail -n 4 $OUT/r05_hostile_fuzz.log | cut -c1-400
