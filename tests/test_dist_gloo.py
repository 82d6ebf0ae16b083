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
rank, local_rank, world = gdist.init_control()
c = parse_file(os.path.join(%r, "tests", "golden", "add64.gcf"))
plan = engine.Plan(c.Gates, c.NumWires, c.num_inputs, c.num_outputs)
total = %d
lo, hi = gdist.shard_range(total, rank, world)
out = np.zeros((hi - lo, c.num_outputs), np.uint8)
for k, i in enumerate(range(lo, hi)):
    bits = np.frombuffer(drbg("dist%%d" %% i, c.num_inputs), np.uint8) & 1
    out[k] = plan.simulate(bits)
tr = gdist.GlooGather(rank, world)
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
