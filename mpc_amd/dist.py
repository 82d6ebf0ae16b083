"""Multi-GPU host logic: independent circuit instances shard embarrassingly across ranks (one process per GPU);
the only exchange is a terminal all-gather of the decoded output bits / output labels.

The data-path collective is gc_comm_allgather of libgcengine.so (ncclAllGather of RCCL over xGMI, called from the
C ABI — what a Go host binds).  torch.distributed appears here only as the launcher's control plane: a gloo (CPU)
group that hands the 128-byte ncclUniqueId from rank 0 to the other ranks — the job a Go host does over its own
p2p.Conn — and, in the CPU tests, stands in as the gather transport so that the sharding / padding / reassembly
logic is exercised with world_size 2 without a GPU."""
import os

import numpy as np


def shard_range(total, rank, world):
    """contiguous instance range [lo, hi) of `rank` when `total` instances are split over `world` ranks"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(total, world):
    """rows every rank contributes to the gather: the largest shard (smaller shards are zero-padded, the
    collective needs equal sizes on all ranks)"""
    return -(-total // world) if world else 0


def reassemble(gathered, total, world):
    """[world, shard_rows, ...] as gathered -> [total, ...] in instance order, padding dropped"""
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(gathered[r][: hi - lo])
    return np.concatenate(parts, axis=0) if parts else gathered.reshape((0,) + gathered.shape[2:])


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_control():
    """CPU control plane of a torch.distributed.run launch (gloo): rendezvous only, no GPU collective"""
    import torch.distributed as dist

    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    return rank, local_rank, world


def exchange_unique_id(make_id, rank, world):
    """rank 0 draws the communicator id (make_id() -> bytes), every rank returns it"""
    if world == 1:
        return make_id()
    import torch.distributed as dist

    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def open_comm(ctx, rank, world):
    """the RCCL communicator of this rank's gc_ctx (gc_comm_init_rank)"""
    from . import engine

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only mode the host driver supports
    uid = exchange_unique_id(engine.comm_unique_id, rank, world)
    return engine.Comm(ctx, uid, world, rank)


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


def gather_sharded(transport, local_rows, total, world):
    """One rank's rows of a [total, ...] result (its shard_range) -> the whole result on every rank.
    transport.allgather_host(padded [shard_rows, ...]) -> [world, shard_rows, ...]"""
    per = shard_rows(total, world)
    local_rows = np.asarray(local_rows)
    pad = np.zeros((per,) + local_rows.shape[1:], local_rows.dtype)
    pad[: len(local_rows)] = local_rows
    return reassemble(transport.allgather_host(pad), total, world)
