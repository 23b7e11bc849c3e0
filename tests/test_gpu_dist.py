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


def _rank_fused(rank, world, port, backend, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    from piplib_amd import dist as pdist, synth
    ndev = torch.cuda.device_count()
    dev = rank % ndev
    torch.cuda.set_device(dev)
    pdist.init(backend, torch.device("cuda", dev))
    batches = [synth.lexmin_batch(1000 + 7919 * g, 10000, NVAR, NI) for g in (0, 1)]
    full = pdist.solve_sharded_fused(batches, NVAR, engine_device=dev)
    if rank == 0:
        q.put([{k: v.cpu().numpy() for k, v in f.items()} for f in full])
    else:
        assert full is None
    pdist.finish()


def test_configs3_size_two_ranks_fused_equals_one_rank():
    """BASELINE configs[3] at its real size: two 10,000-tableau batches of bench.py's workload (batches 0 and 1: the CPU
    oracle's list has no slow-converging tableau in them), each sharded over two ranks, a rank's two shards fused into
    one workspace and one launch sequence (pipamd_batch_load_part, as bench.py's strong-scaling mode does), results
    gathered to rank 0: bit-equal to a one-rank solve of each batch, pivot totals equal to the oracle's."""
    import json
    import torch
    import torch.multiprocessing as mp
    from gpu_common import gpu_batch
    from piplib_amd import synth
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_fused, args=(r, 2, port, backend, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=800)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_screen.json")))["batches"]
    for g in (0, 1):
        one = gpu_batch(synth.lexmin_batch(1000 + 7919 * g, 10000, NVAR, NI), NVAR, 0, 1)
        f = full[g]
        assert f["status"].shape[0] == 10000
        assert (f["status"] == one.status.cpu().numpy()).all()
        assert (f["pivots"] == one.pivots.cpu().numpy()).all() and (f["cuts"] == one.cuts.cpu().numpy()).all()
        assert (f["sol_num"] == one.sol_num.cpu().numpy()).all() and (f["sol_den"] == one.sol_den.cpu().numpy()).all()
        assert rec[str(g)]["slow"] == [] and int(f["pivots"].sum()) == rec[str(g)]["pivots_screened"]
        assert ((f["status"] == 1) | (f["status"] == 2)).all()


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself (a child
    torch.distributed.run, before the parent touches the GPU) and rank 0's line says n_gpus = 2.  On a one-GPU box the
    ranks share the card and the sums go over gloo (PIPAMD_BENCH_BACKEND); with two GPUs it is RCCL."""
    import json
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["PIPAMD_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu",
                        "--no-others", "--no-dense"], env=env, capture_output=True, timeout=850)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "strong"
    assert out["regions_checked"] is True and out["finished_fraction"] == 1.0 and out["value"] > 0
    assert out["oracle_checked_batches"] >= 1
    assert out["other_scaling"]["scaling"] == "weak"


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
