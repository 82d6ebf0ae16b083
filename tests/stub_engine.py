"""A stand-in for mpc_amd.engine with NO device behind it, for ONE purpose: the CPU suite runs bench.py's real rank entry
(argument handling, per-GPU batch of the N > 1 run, communicator hand-over through the id file, StepLoop / run_timed, the
gather bookkeeping, the JSON line and the error path) at world 2 under torch.distributed.run without a GPU
(tests/test_bench_ranks.py; GC_BENCH_ENGINE=tests.stub_engine).  "Garbling" here fills the output slot with a pattern that
depends on rank, step and instance, so that the gather's content can be checked; the collective is gloo.  Test scaffolding:
nothing in the product imports this module, and nothing here computes a label."""
import os
import time

import numpy as np

from mpc_amd import engine as _real

Plan = _real.Plan
EngineError = _real.EngineError
COMM_ID_BYTES = 128


class _Buf:
    def __init__(self, arr):
        self.a = np.ascontiguousarray(arr)
        self.nbytes = self.a.nbytes
        self.shape, self.dtype = self.a.shape, self.a.dtype

    def __add__(self, off):
        return (self, int(off))

    def numpy(self):
        return self.a.copy()

    def close(self):
        pass


class Context:
    def __init__(self, device):
        if os.environ.get("GC_STUB_RANK_FAILS") == os.environ.get("RANK", "0"):  # (a rank whose device is not there)
            raise EngineError(-5, "gc_ctx_create(%d): stub: invalid device ordinal" % device)
        self.device = device

    def random_u8(self, shape, mod, seed=0):
        return _Buf(np.random.default_rng(seed).integers(0, mod, shape, dtype=np.uint8))

    def zeros(self, shape, dtype=np.uint8):
        return _Buf(np.zeros(shape, dtype))

    def sync(self):
        pass

    def capture(self, fn):
        class G:
            def launch(self_inner):
                fn()
        return G()

    def close(self):
        pass


class DeviceCircuit:
    def __init__(self, ctx, circ):
        self.ctx, self.circ = ctx, circ
        self.info = Plan(circ.Gates, circ.NumWires, circ.num_inputs, circ.num_outputs).info

    def close(self):
        pass


class Batch:
    step_counter = 0

    def __init__(self, dc, batch):
        self.dc, self.batch = dc, batch
        self.last_ms, self.last_launches = 0.5, 1

    def set_graph(self, on):
        pass

    def set_schedule(self, s):
        pass

    def garble(self, key, d_rnd):
        time.sleep(0.0002)

    def select_inputs(self, gb, d_bits):
        pass

    def eval(self, key, gb):
        pass

    def decode(self, ev, dst, d_mis):
        buf, off = dst if isinstance(dst, tuple) else (dst, 0)
        nout = self.dc.circ.num_outputs
        flat = buf.a.reshape(-1)
        rank = int(os.environ.get("RANK", "0"))
        Batch.step_counter += 1
        pat = ((np.arange(self.batch * nout) + 3 * rank + Batch.step_counter) & 1).astype(np.uint8)
        flat[off:off + self.batch * nout] = pat

    def close(self):
        pass


def comm_unique_id():
    return bytes([7]) * COMM_ID_BYTES


def comm_version():
    return 0


class Comm:
    """gloo in place of RCCL; GC_STUB_COMM_HANG=<rank>: that rank never joins (the bounded-wait test)"""

    def __init__(self, ctx, uid, nranks, rank):
        assert uid == comm_unique_id()
        if os.environ.get("GC_STUB_COMM_HANG") == str(rank):
            time.sleep(3600)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if nranks > 1 and not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=nranks)
        self.rank, self.nranks = rank, nranks

    def allgather(self, d_send, d_recv, nbytes):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(d_send.a.reshape(-1).view(np.uint8)[:nbytes].copy())
        out = torch.empty(self.nranks * nbytes, dtype=torch.uint8)
        if self.nranks > 1:
            dist.all_gather_into_tensor(out, t)
        else:
            out[:] = t
        d_recv.a.reshape(-1).view(np.uint8)[: self.nranks * nbytes] = out.numpy()

    def allgather_host(self, local):
        a = np.ascontiguousarray(local)
        send, recv = _Buf(a.copy()), _Buf(np.zeros((self.nranks,) + a.shape, a.dtype))
        self.allgather(send, recv, a.nbytes)
        return recv.a

    def allreduce_max(self, v):
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(v)], dtype=torch.float64)
        if self.nranks > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        import torch.distributed as dist
        if self.nranks > 1:
            dist.barrier()

    def close(self):
        import torch.distributed as dist
        if self.nranks > 1 and dist.is_initialized():
            dist.destroy_process_group()
