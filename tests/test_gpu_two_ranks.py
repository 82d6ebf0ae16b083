"""Two rank processes of bench.py on ONE GPU (the only N > 1 this pool allows): the launcher environment, the communicator id
through the file, gc_comm_init_rank with world 2, config 4's shape, the gathers behind the decode on the ctx stream, barrier,
max over ranks, the JSON line.  RCCL itself refuses two ranks on one device ("Duplicate GPU detected",
profiles/r03_two_rank_probe.txt), so the collective library underneath gc_comm_* is tests/standin_rccl (GC_RCCL_PATH): the
same ten entry points over a shared-memory segment — everything above it is the product's own path; RCCL and xGMI are not
part of this test."""
import json
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "standin_rccl", "standin_rccl.cpp")


@pytest.fixture(scope="module")
def standin(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc to build the stand-in collective library")
    so = str(tmp_path_factory.mktemp("standin") / "librccl_standin.so")
    subprocess.check_call([hipcc, "-shared", "-fPIC", "-O2", "-o", so, SRC])
    return so


def _ranks(world, args, standin, tmp_path, timeout=600):
    env = dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
               TORCHELASTIC_RUN_ID="standin%d" % os.getpid(), GC_RENDEZVOUS_DIR=str(tmp_path), GC_RCCL_PATH=standin,
               GC_BENCH_DEVICE="0", TMPDIR=str(tmp_path))  # (TMPDIR: where the stand-in keeps its shared file)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + args,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    return procs, outs


def test_two_ranks_on_one_gpu_run_the_whole_path(standin, tmp_path):
    procs, outs = _ranks(2, ["--steps", "11", "--warmup", "3", "--batch", "1024", "--no-cpu-baseline"], standin, tmp_path)
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s" % (r, err[-3000:])
    lines = [json.loads(l) for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")], "rank 0 prints the one line"
    j = lines[0]
    assert "error" not in j and j["n_gpus"] == 2 and j["n_ranks_seen"] == 2 and j["scaling"] == "weak"
    assert j["steps"] == 11 and j["warmup"] == 3 and j["config"]["instances_per_gpu"] == 1024
    c4 = j["config4"]
    assert c4["instances_total"] == 2048 and c4["gathered_outputs_ok"] and c4["gathered_bytes_per_gpu"] == 8 * 1024 * 128
    assert j["config"]["gathers"] == c4["gathers_in_timed_region"] == 1 + 2
    assert j["config"]["outputs_ok"] and j["value"] > 0
    assert abs(j["value"] - 6400 * 1024 * 2 * 11 / (j["ms_per_step"] * 11e-3)) < 1e-6 * j["value"]
    assert not [f for f in os.listdir(tmp_path) if f.startswith("gc_comm_id.")], "rank 0 removes the id file"


def test_a_rank_that_never_arrives_ends_the_other_with_the_error_line(standin, tmp_path):
    """world 2 announced, one process started: gc_comm_init_rank cannot complete; the watchdog ends rank 0 with ONE JSON
    error line and a non-zero exit inside the init timeout"""
    env = dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29518",
               TORCHELASTIC_RUN_ID="alone%d" % os.getpid(), GC_RENDEZVOUS_DIR=str(tmp_path), GC_RCCL_PATH=standin, GC_BENCH_DEVICE="0",
               TMPDIR=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "256",
                        "--no-cpu-baseline", "--init-timeout", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0].get("error") and lines[0]["value"] is None and "communicator" in lines[0]["stage"], r.stdout[-1500:]
