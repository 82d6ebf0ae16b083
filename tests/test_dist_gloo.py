"""N>1 path on CPU: two processes over gloo shard a batch of instances, each decodes its shard (with the
plaintext evaluator standing in for the GPU), and the single all-gather reassembles the batch."""
import os
import subprocess
import sys

import numpy as np

from mpc_amd.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import oracle
from mpc_amd import dist as gdist, parse_file
from tests.util import drbg
rank, local_rank, world = gdist.init("gloo")
c = parse_file(os.path.join(%r, "tests", "golden", "add64.gcf"))
total = 10
per = total // world
lo, hi = gdist.shard_range(total, rank, world)
assert hi - lo == per
out = np.zeros((per, c.num_outputs), np.uint8)
for k, i in enumerate(range(lo, hi)):
    bits = np.frombuffer(drbg("dist%%d" %% i, c.num_inputs), np.uint8) & 1
    out[k] = oracle.compute(c.Gates, c.NumWires, c.num_inputs, bits)[-c.num_outputs:]
allout = gdist.gather_outputs(torch.from_numpy(out), world)
t = gdist.max_over_ranks(0.5 + rank, world)
if rank == 0:
    np.save(sys.argv[1], allout.reshape(total, -1).numpy())
    assert t == 0.5 + world - 1
import torch.distributed as dist
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 8, 1024, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_ranks_gloo(tmp_path, add64_circ):
    import oracle
    from tests.util import drbg

    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    out = tmp_path / "all.npy"
    port = 29500 + os.getpid() % 2000
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    c = add64_circ
    for i in range(10):
        bits = np.frombuffer(drbg("dist%d" % i, c.num_inputs), np.uint8) & 1
        want = oracle.compute(c.Gates, c.NumWires, c.num_inputs, bits)[-c.num_outputs:]
        assert (got[i] == want).all()
