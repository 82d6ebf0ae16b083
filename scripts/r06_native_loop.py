"""round 6: BENCH_r05's native_host.big130 failure in the setting it happened in — a parent interpreter that holds a ctx of its
own (bench.py's) and has just run the Ed25519-shaped program, then tools/stream_driver as a child — N times, the child's
stderr kept.  usage: r06_native_loop.py [N]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_stream as bs  # noqa: E402
from mpc_amd import engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = engine.Context(0)
key = bytes(range(32))
for i in range(n):
    t0 = time.time()
    r = bs.run_program("ed25519like", key, ctx, window=bs.WINDOWS["ed25519like"], view=True)
    t1 = time.time()
    try:
        steps, prim = bs.PROGRAMS["big130"]()
        nat = bs.run_native_steps(steps, prim, bs.stream_rnd("big130", len(prim)), key, 2)
        out = {"i": i, "ok": nat["sha256"] == bs.golden_sha("big130", key), "garble_s": nat["garble_s"], "eval_blocks_pinned_s": nat["eval_blocks_pinned_s"],
               "coop_state": nat["coop_state"], "coop_timeouts": nat["coop_timeouts"]}
    except Exception as e:
        out = {"i": i, "ok": False, "error": str(e)[:1500]}
    out["parent_s"] = round(t1 - t0, 2)
    out["child_s"] = round(time.time() - t1, 2)
    print(json.dumps(out), flush=True)
