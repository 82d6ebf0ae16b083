#!/bin/bash
# the GPU suite and the default bench line several times over on one box: does anything fail once in a while?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out/r06c_soak.txt; : > $OUT
for i in 1 2; do
  timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2 >> $OUT
done
for i in 1 2; do
  t0=$SECONDS
  timeout 600 python bench.py > /tmp/b$i.json 2> /tmp/b$i.err; rc=$?
  echo "bench $i rc=$rc wall $((SECONDS - t0)) s" >> $OUT
  python - /tmp/b$i.json >> $OUT 2>&1 <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = j["stream"]
print("value %.4g traffic %s config3 %.4g ssa23 %.3g mixed %.3g ed %.3g" % (j["value"], j["roofline"]["traffic"], j["config3"]["and_gates_per_s"],
      s["ssa23"]["garble_gates_per_s"], s["mixed"]["garble_gates_per_s"], s["ed25519like"]["garble_gates_per_s"]))
print("native_host ok:", {k: v.get("sha256_ok", v.get("error", "?")) for k, v in s["native_host"].items()}, "first_attempt_errors:",
      [k for k, v in s["native_host"].items() if "first_attempt_error" in v])
print("stream sha ok:", all(s[k]["sha256_ok"] for k in ("ed25519like", "ssa23", "mixed", "uniform512", "uniform4096")), "side wall", j["side_rows_wall_s"])
PY
done
cat $OUT
