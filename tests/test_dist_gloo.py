"""N>1 host logic on CPU: two processes (torch.distributed.run, gloo) own unequal shards of a batch of instances,
each walks ITS instances through the engine's flattened unit program on plaintext bits (gc_plan_simulate: the exact
unit / slot / store program the fused kernels execute — product code, no GPU needed), and the sharded gather
(padding of the smaller shard, reassembly in instance order, id hand-over from rank 0, max-over-ranks timing)
reassembles the batch.  The RCCL transport of the same calls (gc_comm_*) runs in tests/test_gpu_config4.py."""
import os
import subprocess
import sys

import numpy as np

from mpc_amd.dist import reassemble, shard_range, shard_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL = 11  # not a multiple of the world size: the last rank's shard is padded for the collective

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from mpc_amd import dist as gdist, engine, parse_file
from tests.util import drbg
from tests.gloo_transport import GlooGather
rank, local_rank, world = gdist.init_control()
c = parse_file(os.path.join(%r, "tests", "golden", "add64.gcf"))
plan = engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs)
total = %d
lo, hi = gdist.shard_range(total, rank, world)
out = np.zeros((hi - lo, c.num_outputs), np.uint8)
for k, i in enumerate(range(lo, hi)):
    bits = np.frombuffer(drbg("dist%%d" %% i, c.num_inputs), np.uint8) & 1
    out[k] = plan.simulate(bits)
tr = GlooGather(rank, world)
uid = gdist.exchange_unique_id(lambda: b"id-from-rank-%%d" %% rank, rank, world)
assert uid == b"id-from-rank-0"
allout = gdist.gather_sharded(tr, out, total, world)
t = tr.allreduce_max(0.5 + rank)
tr.barrier()
if rank == 0:
    np.save(sys.argv[1], allout)
    assert t == 0.5 + world - 1
import torch.distributed as dist
dist.destroy_process_group()
'''


# bench.py's step loop (mpc_amd/dist.py: StepLoop + run_timed) with a numpy accumulator standing in for the device
# buffer and the gloo transport standing in for gc_comm_allgather: every step's outputs must be gathered exactly once,
# also when the number of steps is not a multiple of the accumulator's slots (the tail is flushed inside the timed
# region), on every rank, and the reported time is the slowest rank's.  The communicator id travels through the file
# exchange bench.py uses (no torch involved in that part).
LOOP_WORKER = r'''
import os, sys, json, time
sys.path.insert(0, %r)
import numpy as np
from mpc_amd import dist as gdist
from tests.gloo_transport import GlooGather
rank, local_rank, world = gdist.init_control()   # gloo: only the stand-in transport needs it
steps, warmup, K = int(sys.argv[2]), int(sys.argv[3]), 8
uid = gdist.exchange_unique_id_file(lambda: b"file-id-from-rank-%%d" %% rank + bytes(100), rank, world,
                                    directory=os.path.dirname(sys.argv[1]))
assert uid.startswith(b"file-id-from-rank-0")
tr = GlooGather(rank, world)
acc = np.zeros((K, 4), np.int64)
seen = {}          # (rank, step) -> times gathered, as observed by THIS rank
counter = [0]
def launch(j):
    acc[j] = (rank, counter[0], j, 7)   # "outputs" of step counter[0] of this rank land in slot j
    counter[0] += 1
def gather(nfresh):
    allacc = tr.allgather_host(acc)     # [world, K, 4], the whole accumulator as bench.py gathers it
    for r in range(world):
        for j in range(nfresh):
            assert allacc[r, j, 0] == r and allacc[r, j, 2] == j and allacc[r, j, 3] == 7
            key = (r, int(allacc[r, j, 1]))
            seen[key] = seen.get(key, 0) + 1
fences = [0]
def fence():
    fences[0] += 1
    tr.barrier()
loop = gdist.StepLoop(K, launch, gather)
clock_vals = iter([10.0, 10.0 + 1.0 + rank])   # rank r "takes" 1 + r seconds
elapsed = gdist.run_timed(loop, fence, steps, warmup, allreduce_max=tr.allreduce_max, clock=lambda: next(clock_vals))
assert elapsed == 1.0 + (world - 1), elapsed
assert fences[0] == 2 and loop.steps_done == loop.steps_gathered == steps + warmup == counter[0]
want_gathers = -(-warmup // K) + -(-steps // K)
assert loop.gathers == want_gathers, (loop.gathers, want_gathers)
for r in range(world):
    for i in range(steps + warmup):
        assert seen.get((r, i)) == 1, (r, i, seen.get((r, i)))
assert len(seen) == world * (steps + warmup)
# one rank, no collective: the loop degenerates to launch(0) per step
solo = gdist.StepLoop(K, lambda j: None, None)
gdist.run_timed(solo, lambda: None, 5, 2)
assert solo.slots == 1 and solo.steps_done == 7 and solo.gathers == 0
gdist.cleanup_unique_id_file(rank, directory=os.path.dirname(sys.argv[1]))
if rank == 0:
    open(sys.argv[1], "w").write("ok %%d" %% world)
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 8, 1024, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
            assert shard_rows(total, world) == max(h - l for l, h in spans)


def test_reassemble_drops_padding():
    total, world = 7, 3
    per = shard_rows(total, world)
    g = np.full((world, per, 2), 255, np.uint8)
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        for k, i in enumerate(range(lo, hi)):
            g[r, k] = i
    out = reassemble(g, total, world)
    assert out.shape == (7, 2) and (out[:, 0] == np.arange(7)).all()


def test_config4_shape():
    # 65 536 instances over 8 ranks: 8 192 each, no padding
    assert [shard_range(65536, r, 8) for r in (0, 7)] == [(0, 8192), (57344, 65536)] and shard_rows(65536, 8) == 8192


import pytest


@pytest.mark.parametrize("world,steps,warmup", [(2, 21, 3), (3, 13, 9), (2, 16, 8), (3, 1, 0)])
def test_bench_step_loop_gathers_every_step_once(tmp_path, world, steps, warmup):
    script = tmp_path / "loop_worker.py"
    script.write_text(LOOP_WORKER % (ROOT,))
    out = tmp_path / "done.txt"
    port = 31500 + (os.getpid() * 7 + world * 13 + steps) % 2000
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script), str(out), str(steps), str(warmup)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert out.read_text() == "ok %d" % world
    assert not [f for f in os.listdir(tmp_path) if f.startswith("gc_comm_id.")], "rank 0 removes the id file"


def test_unique_id_file_single_rank_and_timeout(tmp_path):
    from mpc_amd import dist as gdist
    assert gdist.exchange_unique_id_file(lambda: b"x" * 128, 0, 1) == b"x" * 128
    with pytest.raises(TimeoutError):
        gdist.exchange_unique_id_file(lambda: b"", 1, 2, timeout=0.2, directory=str(tmp_path))


def test_comm_argument_errors_need_no_gpu():
    """gc_comm_init_rank / gc_comm_* reject bad arguments before touching RCCL or a device"""
    import ctypes as C
    from mpc_amd import engine
    L = engine.lib()
    st = C.c_int(0)
    uid = (C.c_uint8 * 128)()
    assert not L.gc_comm_init_rank(None, uid, 128, 2, 0, C.byref(st)) and st.value == engine.GC_E_ARG
    assert L.gc_comm_get_unique_id(None, 128) == engine.GC_E_ARG
    assert L.gc_comm_get_unique_id(uid, 64) == engine.GC_E_ARG
    assert L.gc_comm_allgather(None, None, None, 16) == engine.GC_E_ARG
    assert L.gc_comm_barrier(None) == engine.GC_E_ARG
    assert L.gc_comm_rank(None) == -1 and L.gc_comm_nranks(None) == 0
    outs = (C.c_void_p * 2)()
    assert L.gc_comm_init_all(None, 2, outs) == engine.GC_E_ARG
    v = C.c_double(1.0)
    assert L.gc_comm_allreduce_max(None, C.byref(v)) == engine.GC_E_ARG
    L.gc_comm_destroy(None)  # no-op


def test_two_ranks_gloo(tmp_path, add64_circ):
    import oracle
    from tests.util import drbg

    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT, TOTAL))
    out = tmp_path / "all.npy"
    port = 29500 + os.getpid() % 2000
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    c = add64_circ
    assert got.shape == (TOTAL, c.num_outputs)
    for i in range(TOTAL):
        bits = np.frombuffer(drbg("dist%d" % i, c.num_inputs), np.uint8) & 1
        want = oracle.compute(c.Gates, c.NumWires, c.num_inputs, bits)[-c.num_outputs:]
        assert (got[i] == want).all()
