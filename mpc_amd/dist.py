"""Multi-GPU plumbing: independent circuit instances shard embarrassingly across ranks (one process per
GPU); the only exchange is a terminal all-gather of the decoded output bits / output labels
(SURVEY.md §8e).  torch.distributed is used as plumbing only ("nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests) — the data path has no collective."""
import os


def shard_range(total, rank, world):
    """contiguous instance range [lo, hi) of `rank` when `total` instances are split over `world` ranks"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend):
    import torch.distributed as dist

    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl":
            import torch
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def gather_outputs(local, world):
    """all-gather equal-sized per-rank result tensors into [world, ...] (the single collective)"""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local.unsqueeze(0)
    shape = tuple(local.shape)
    out = torch.empty((world * shape[0],) + shape[1:], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())  # concatenation along dim 0 (gloo and nccl agree)
    return out.view((world,) + shape)


def max_over_ranks(value, world, device="cpu"):
    import torch
    import torch.distributed as dist

    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
