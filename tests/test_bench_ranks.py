"""bench.py's real rank entry at world 2 on CPU: `python -m torch.distributed.run --nproc-per-node 2
bench.py --gpus 2 ...` with tests/stub_engine.py in place of the device (GC_BENCH_ENGINE) — the communicator id hand-over
through the file, config 4's per-GPU batch, StepLoop / run_timed, the gather and its bookkeeping, the fields of the JSON
line; and the failure path: a rank that never joins ends the run with ONE JSON error line and a non-zero exit inside the
init timeout instead of hanging."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, extra, env_extra, tmp_path, timeout=240):
    import socket
    with socket.socket() as sk:  # a port nobody holds right now (the launcher's rendezvous)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT, GC_BENCH_ENGINE="tests.stub_engine", GC_RENDEZVOUS_DIR=str(tmp_path), **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "19",
           "--warmup", "5", "--no-cpu-baseline"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_bench_two_ranks_runs_config4_shape(tmp_path):
    r = _launch(2, [], {}, tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    j = lines[0]
    assert "error" not in j
    assert j["n_gpus"] == 2 and j["n_ranks_seen"] == 2 and j["steps"] == 19 and j["warmup"] == 5 and j["scaling"] == "weak"
    assert j["config"]["instances_per_gpu"] == 8192 and "batch=8192" in j["config"]["workload"]   # config 4's share per GPU
    c4 = j["config4"]
    assert c4["instances_total"] == 16384 and c4["gathered_bytes_per_gpu"] == 8 * 8192 * 128 and c4["gathered_outputs_ok"]
    assert c4["gathers_in_timed_region"] == j["config"]["gathers"] == -(-5 // 8) + -(-19 // 8)
    assert j["value"] > 0 and abs(j["value"] - 6400 * 8192 * 2 * 19 / (j["ms_per_step"] * 19e-3)) < 1e-6 * j["value"]
    assert not [f for f in os.listdir(tmp_path) if f.startswith("gc_comm_id.")], "rank 0 removes the id file"
    # round 6: every rank's own time and rate over the timed steps beside the job's (the slowest rank's time, all ranks' units)
    pr = j["per_rank"]
    assert len(pr["value"]) == len(pr["elapsed_s"]) == 2 and pr["min"] <= pr["max"] and min(pr["elapsed_s"]) > 0
    assert abs(pr["min"] - 6400 * 8192 * 19 / max(pr["elapsed_s"])) < 1e-6 * pr["min"]
    assert j["value"] <= sum(pr["value"]) * (1 + 1e-9)  # (the job's clock is the slowest rank's)


def test_bench_explicit_batch_and_single_rank_default(tmp_path):
    r = _launch(2, ["--batch", "96"], {}, tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_lines(r.stdout)[0]
    assert j["config"]["instances_per_gpu"] == 96 and j["config4"]["instances_total"] == 192
    # N = 1 keeps BASELINE config 2 (the line the driver's BENCH file holds)
    env = dict(os.environ, PYTHONPATH=ROOT, GC_BENCH_ENGINE="tests.stub_engine")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                         "--no-iknp", "--no-host-api", "--no-stream", "--no-config3", "--no-synthetic", "--no-extra-rows"],
                        env=env, capture_output=True, text=True, timeout=240)
    assert r1.returncode == 0, r1.stderr[-3000:]
    j1 = _json_lines(r1.stdout)[0]
    assert j1["n_gpus"] == 1 and j1["config"]["instances_per_gpu"] == 1024 and "config4" not in j1 and j1["n_ranks_seen"] == 1


def test_bench_missing_rank_ends_with_one_error_line(tmp_path):
    t0 = time.time()
    r = _launch(2, ["--init-timeout", "6"], {"GC_STUB_COMM_HANG": "1"}, tmp_path, timeout=120)
    assert time.time() - t0 < 90
    assert r.returncode != 0
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0].get("error") and lines[0]["value"] is None and lines[0]["n_gpus"] == 2, r.stdout[-2000:]
    assert "communicator" in lines[0]["stage"]


def test_bench_rank_that_dies_early_still_leaves_one_error_line(tmp_path):
    """rank 1 cannot open its device (what a 2-rank launch on a 1-GPU box does): it fails at once, the launcher sends SIGTERM
    to rank 0 — which may be inside a C call —, and rank 0 still prints its ONE error line before it leaves"""
    t0 = time.time()
    r = _launch(2, ["--init-timeout", "30"], {"GC_STUB_RANK_FAILS": "1"}, tmp_path, timeout=120)
    assert time.time() - t0 < 60 and r.returncode != 0
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0].get("error") and lines[0]["n_gpus"] == 2, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rank 1 failed at stage 'context'" in r.stderr
