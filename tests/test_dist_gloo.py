"""N > 1 path on CPU: two gloo ranks, each owning its own shard of tableaux (no data-path
collective), totals gathered with all_reduce -- the same piplib_amd.dist code bench.py runs
over RCCL.  The per-rank "solve" here is the CPU oracle (test infrastructure)."""
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


def _solve_shard(seed, n):
    from piplib_amd import synth
    rows = synth.lexmin_batch(seed, n, 6, 8, nnz=3, cmax=3, x0max=5)
    probs = [synth.Problem(6, 0, 8, 0, -1, 1, rows[b], np.zeros((0, 1), np.int64)) for b in range(n)]
    out = pb.run_batch(pb.ORACLEPIP, probs, pb.F_NOSIMPLIFY)
    return out.total_pivots, sum(1 for r in out.results if r.status == pb.ST_OK)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from piplib_amd import dist as pdist
    r, w, _ = pdist.init("gloo")
    assert (r, w) == (rank, world)
    piv, ok = _solve_shard(pdist.shard_seed(500, rank), 12)
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
    want = [_solve_shard(500 + r, 12) for r in range(2)]
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
