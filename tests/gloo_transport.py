"""Stand-in transport of the CPU tests (world_size 2 and 3 under torch.distributed.run, gloo): the call shape of
engine.Comm for host arrays.  Test scaffolding — the product's transport is gc_comm_* (RCCL) behind the C ABI."""
import numpy as np


class GlooGather:
    """stand-in transport of the CPU tests: same call shape as engine.Comm for host arrays"""

    def __init__(self, rank, world):
        self.rank, self.nranks = rank, world

    def allgather_host(self, local):
        import torch
        import torch.distributed as dist

        t = torch.from_numpy(np.ascontiguousarray(local))
        if self.nranks == 1:
            return t.unsqueeze(0).numpy()
        out = torch.empty((self.nranks,) + tuple(t.shape), dtype=t.dtype)
        dist.all_gather_into_tensor(out.view(self.nranks * t.shape[0], *t.shape[1:]), t)
        return out.numpy()

    def allreduce_max(self, value):
        import torch
        import torch.distributed as dist

        if self.nranks == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        import torch.distributed as dist

        if self.nranks > 1:
            dist.barrier()
