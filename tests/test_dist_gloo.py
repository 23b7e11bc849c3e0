"""N > 1 path on CPU: two gloo ranks.  (1) the final gather of piplib_amd.dist.gather_results on
the real shapes of BASELINE configs[3] (10,000 tableaux, 127 unknowns; uneven slices; the per-rank
payload is a deterministic function of the tableau index, so order and padding are checked
exactly); (2) the totals bench.py sums with all_reduce.  The HIP path cannot run here: the same
code with the engine behind it runs under -m gpu (tests/test_gpu_dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import pipbatch as pb


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_totals(seed, n):
    """stand-in for a rank's (pivots, finished tableaux): the sums are what is under test here"""
    return 1000 * seed + n, n - seed % 3


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from piplib_amd import dist as pdist
    r, w, _ = pdist.init("gloo")
    assert (r, w) == (rank, world)
    piv, ok = _shard_totals(pdist.shard_seed(500, rank), 12)
    pdist.barrier()
    tot, tmax = pdist.gather_totals([piv, 12, ok], 1.0 + rank)
    if rank == 0:
        q.put((tot, tmax))
    pdist.finish()


@pytest.mark.timeout(300)
def test_two_rank_weak_scaling_totals():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    tot, tmax = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [_shard_totals(500 + r, 12) for r in range(2)]
    assert tot[0] == sum(w[0] for w in want)      # pivots summed over ranks
    assert tot[1] == 24                           # tableaux
    assert tot[2] == sum(w[1] for w in want)
    assert tmax == 2.0                            # max over ranks


def test_shard_range_partitions():
    from piplib_amd import dist as pdist
    for total in (0, 1, 7, 10000, 10001):
        for world in (1, 2, 3, 8):
            spans = [pdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _gather_worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from piplib_amd import dist as pdist
    pdist.init("gloo")
    lo, hi = pdist.shard_range(total, rank, world)
    idx = torch.arange(lo, hi, dtype=torch.int64)
    parts = {"status": (idx % 3 + 1).to(torch.int32), "pivots": (idx * 7 % 251).to(torch.int32),
             "cuts": (idx % 17).to(torch.int32),
             "sol_num": (idx[:, None, None] * 1000 + torch.arange(127)[None, :, None]).to(torch.int64),
             "sol_den": (idx[:, None] + torch.arange(127)[None, :] + 1).to(torch.int64)}
    full = pdist.gather_results(parts, total)
    if rank == 0:
        q.put({k: v.numpy() for k, v in full.items()})
    else:
        assert full is None
    pdist.finish()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("total,world", [(10000, 2), (10001, 2), (5, 3)])
def test_final_gather_real_shapes(total, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    idx = np.arange(total, dtype=np.int64)
    assert (full["status"] == idx % 3 + 1).all() and (full["pivots"] == idx * 7 % 251).all()
    assert (full["cuts"] == idx % 17).all()
    assert full["sol_num"].shape == (total, 127, 1) and full["sol_den"].shape == (total, 127)
    assert (full["sol_num"][:, :, 0] == idx[:, None] * 1000 + np.arange(127)[None, :]).all()
    assert (full["sol_den"] == idx[:, None] + np.arange(127)[None, :] + 1).all()
