"""-m gpu: the multi-rank path with the HIP engine behind it.

Two ranks (one process each) shard one batch with piplib_amd.dist.solve_sharded, solve their slices
through the C ABI and gather every result to rank 0; the gathered results must equal a 1-rank run
bit for bit.  With two or more GPUs visible the ranks use one GPU each over RCCL ("nccl"); on a
one-GPU box both ranks share the card and the gather goes over gloo (same code path up to the
backend).  Also: two engines in one process (second device when there is one), each on tableaux
whose LDS image needs the > 48 KiB opt-in, which is per device."""
import os
import socket

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

NVAR, NI, TOTAL = 127, 64, 600


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, backend, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    from piplib_amd import dist as pdist, synth
    ndev = torch.cuda.device_count()
    dev = rank % ndev
    torch.cuda.set_device(dev)
    pdist.init(backend, torch.device("cuda", dev))
    rows = synth.lexmin_batch(4242, TOTAL, NVAR, NI)
    full = pdist.solve_sharded(rows, NVAR, engine_device=dev)
    if rank == 0:
        q.put({k: v.cpu().numpy() for k, v in full.items()})
    else:
        assert full is None
    pdist.finish()


def test_two_ranks_gather_equals_one_rank():
    import torch
    import torch.multiprocessing as mp
    from gpu_common import gpu_batch
    from piplib_amd import synth
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, backend, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=600)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    g = gpu_batch(synth.lexmin_batch(4242, TOTAL, NVAR, NI), NVAR, 0, 1)
    assert (full["status"] == g.status.cpu().numpy()).all()
    assert (full["pivots"] == g.pivots.cpu().numpy()).all() and (full["cuts"] == g.cuts.cpu().numpy()).all()
    assert (full["sol_num"] == g.sol_num.cpu().numpy()).all() and (full["sol_den"] == g.sol_den.cpu().numpy()).all()
    assert full["pivots"].sum() > 0 and (full["status"] > 0).all()


def test_two_engines_one_process_large_lds():
    """300-row tableaux of 510 unknowns need a ~60 KiB LDS image (opt-in above 48 KiB, a per-device
    function attribute): two engines of one process, on two devices when there are two, both run
    them and agree with the oracle."""
    import torch
    from gpu_common import oracle_batch, solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    rows = synth.lexmin_batch(61, 4, 510, 300, nnz=3, cmax=3, x0max=4)
    o = oracle_batch(rows, 510, 0, 1)
    devs = [0, 1 if torch.cuda.device_count() >= 2 else 0]
    batches = []
    for d in devs:
        with torch.cuda.device(d):
            e = eng.Engine(d)
            b = eng.Batch(e, rows, 510, 0, tflags=eng.T_INT, cap_cuts=200)
            b.load()
            b.solve()
            b.fetch()
            torch.cuda.synchronize(d)
            batches.append(b)
    for b in batches:
        st, pv = b.status.cpu().numpy(), b.pivots.cpu().numpy()
        num, den = b.sol_num.cpu().numpy(), b.sol_den.cpu().numpy()
        for i, r in enumerate(o.results):
            assert pv[i] == r.pivots
            got = "()" if st[i] == eng.ST_NIL else pb.squash(solution_text(num[i], den[i]))
            assert got == pb.squash(r.text)
