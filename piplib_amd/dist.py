"""Multi-GPU plumbing: one process per GPU, independent tableaux per rank -- the pivot path has no
collective.  torch.distributed (backend "nccl" = RCCL over xGMI on GPUs, gloo in CPU tests) is used
for exactly one exchange: the final gather of the results to rank 0 (`solve_sharded`), plus the
sums bench.py reports (`gather_totals`).

    rows (one 10k-tableau batch, the same on every rank or only its own slice)
      -> shard_range(total, rank, world)          contiguous slice per rank, sizes differ by <= 1
      -> the rank's own GPU: load + traiter() for its slice (piplib_amd.engine.Batch)
      -> gather_results(...)                      status / pivots / cuts / solutions, input order, on rank 0
"""
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


def gather_results(parts, total, device="cpu", dst=0):
    """The final gather (BASELINE configs[3]: "RCCL over xGMI only for the final gather").

    `parts`: this rank's results for its shard_range slice, a dict of tensors whose first dimension
    is the slice length (status, pivots, cuts int32; sol_num, sol_den int64 with any trailing
    shape).  Returns, on rank `dst`, the dict of full-length tensors in input order (None on the
    other ranks).  Slices differ in length by at most one tableau, so every rank pads its tensors
    to the longest slice and ONE `gather` per tensor moves them to `dst` -- only `dst` receives
    (world - 1 slices over its xGMI links; an all_gather would deliver every slice to every rank,
    world times the traffic); a 1-rank job returns its parts unchanged."""
    import torch
    if not _active():
        return parts
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    spans = [shard_range(total, r, world) for r in range(world)]
    longest = max(hi - lo for lo, hi in spans)
    out = {} if rank == dst else None
    for key in sorted(parts):
        t = parts[key].to(device)
        lo, hi = spans[rank]
        assert t.shape[0] == hi - lo, (key, t.shape, lo, hi)
        pad = torch.zeros((longest,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:hi - lo] = t
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst)
        if rank == dst:
            out[key] = torch.cat([bufs[r][:spans[r][1] - spans[r][0]] for r in range(world)], dim=0)
    return out


def solve_sharded(rows, nvar, nparm=0, tflags=1, entier_bits=64, device=None, engine_device=None):
    """One batch over all ranks: every rank holds `rows` (the whole batch, host or device), solves its
    shard_range slice on its own GPU (HIP path, no CPU fallback) and rank 0 receives every
    tableau's status, pivot and cut counts and solution in input order.  Returns that dict on rank
    0, None elsewhere."""
    import torch
    from piplib_amd import engine as eng
    rank, world, local = env_rank()
    if engine_device is None:
        engine_device = local % max(1, torch.cuda.device_count())
    total = int(rows.shape[0])
    lo, hi = shard_range(total, rank, world)
    dev = torch.device("cuda", engine_device)
    mine = torch.as_tensor(rows[lo:hi], dtype=torch.int64).to(dev)
    parts = None
    with torch.cuda.device(dev):
        if hi > lo:
            e = eng.Engine(engine_device)
            b = eng.Batch(e, mine, nvar, nparm, tflags=tflags, entier_bits=entier_bits)
            b.load()
            b.solve()
            b.fetch()
            torch.cuda.synchronize(dev)
            parts = {"status": b.status, "pivots": b.pivots, "cuts": b.cuts, "sol_num": b.sol_num, "sol_den": b.sol_den}
        else:  # more ranks than tableaux: an empty slice still takes part in the gather
            ew = (2,) if entier_bits == 128 else ()
            parts = {"status": torch.zeros(0, dtype=torch.int32, device=dev),
                     "pivots": torch.zeros(0, dtype=torch.int32, device=dev),
                     "cuts": torch.zeros(0, dtype=torch.int32, device=dev),
                     "sol_num": torch.zeros((0, nvar, nparm + 1) + ew, dtype=torch.int64, device=dev),
                     "sol_den": torch.zeros((0, nvar) + ew, dtype=torch.int64, device=dev)}
    # RCCL gathers device tensors; a gloo group (CPU rehearsal with ranks sharing a GPU) host copies
    import torch.distributed as dist
    gdev = dev if (device is None and _active() and dist.get_backend() == "nccl") else (device or "cpu")
    return gather_results(parts, total, gdev)


def solve_sharded_fused(batches, nvar, nparm=0, tflags=1, entier_bits=64, device=None, engine_device=None):
    """Several batches over all ranks with ONE launch sequence per rank (BASELINE configs[3] as bench.py runs it): every
    rank loads its shard_range slice of every batch into one workspace (pipamd_batch_load_part, one call per shard),
    solves them together, and every batch's results are gathered to rank 0 in input order.  `batches`: a list of row
    arrays of the same (ni, ncol).  Returns the list of result dicts on rank 0, None elsewhere."""
    import torch
    from piplib_amd import engine as eng
    rank, world, local = env_rank()
    if engine_device is None:
        engine_device = local % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", engine_device)
    spans = [shard_range(int(r.shape[0]), rank, world) for r in batches]
    parts = [torch.as_tensor(r[lo:hi], dtype=torch.int64).to(dev).contiguous() for r, (lo, hi) in zip(batches, spans)]
    n = sum(hi - lo for lo, hi in spans)
    ni, ncol = int(batches[0].shape[1]), int(batches[0].shape[2])
    ew = (2,) if entier_bits == 128 else ()
    with torch.cuda.device(dev):
        if n > 0:
            e = eng.Engine(engine_device)
            b = eng.Batch(e, None, nvar, nparm, tflags=tflags, entier_bits=entier_bits, shape=(n, ni, ncol))
            b.load_parts([p for p in parts if p.shape[0]])
            b.solve()
            b.fetch()
            torch.cuda.synchronize(dev)
            res = {"status": b.status, "pivots": b.pivots, "cuts": b.cuts, "sol_num": b.sol_num, "sol_den": b.sol_den}
        else:
            res = {"status": torch.zeros(0, dtype=torch.int32, device=dev), "pivots": torch.zeros(0, dtype=torch.int32, device=dev),
                   "cuts": torch.zeros(0, dtype=torch.int32, device=dev),
                   "sol_num": torch.zeros((0, nvar, nparm + 1) + ew, dtype=torch.int64, device=dev),
                   "sol_den": torch.zeros((0, nvar) + ew, dtype=torch.int64, device=dev)}
    import torch.distributed as dist
    gdev = dev if (device is None and _active() and dist.get_backend() == "nccl") else (device or "cpu")
    out, off = [], 0
    for r, (lo, hi) in zip(batches, spans):
        mine = {k: v[off:off + hi - lo] for k, v in res.items()}
        off += hi - lo
        out.append(gather_results(mine, int(r.shape[0]), gdev))
    return out if rank == 0 or not _active() else None


def _active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def world_size():
    """ranks of the initialised process group (1 without one)"""
    if _active():
        import torch.distributed as dist
        return dist.get_world_size()
    return 1


def barrier():
    if _active():
        import torch.distributed as dist
        dist.barrier()


def finish():
    if _active():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
