"""Multi-GPU plumbing: one process per GPU, independent tableaux per rank (no data-path
collective); torch.distributed (RCCL on GPUs, gloo in CPU tests) only gathers the totals."""
import os


def env_rank():
    """(rank, world_size, local_rank) from the torchrun environment (1-process default)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend, device=None):
    """Initialise the default process group when WORLD_SIZE > 1; returns (rank, world, local)."""
    rank, world, local = env_rank()
    if world > 1 or "RANK" in os.environ:  # under torchrun also for a single rank: same code path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_seed(base_seed, rank):
    """Weak scaling: every rank draws its own batch of the same shape from its own seed."""
    return base_seed + rank


def shard_range(total, rank, world):
    """Strong scaling helper: contiguous [lo, hi) slice of `total` tableaux for this rank."""
    per, rem = divmod(total, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def gather_totals(counts, seconds, device="cpu"):
    """Sum `counts` (list of numbers) and take the max of `seconds` over all ranks."""
    import torch
    tot = torch.tensor([float(c) for c in counts], dtype=torch.float64, device=device)
    tmax = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if _active():
        import torch.distributed as dist
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return tot.cpu().tolist(), float(tmax.item())


def _active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def barrier():
    if _active():
        import torch.distributed as dist
        dist.barrier()


def finish():
    if _active():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
