#!/usr/bin/env python3
"""bench.py -- pivots/s of the HIP PIP engine on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the one the metric is quoted on): per GPU a batch of
10,000 synthetic 64x128 int64 tableaux (nvar = 127 unknowns, 64 inequality rows, constant
column; no parameters), integer solve with Gomory cuts, one workgroup per tableau.
A "step" = tab_get-style load of the batch into the HBM row store + the whole traiter()
pivot loop for every tableau (inputs are resident in HBM before the timed region).
Steps are pipelined: up to --pipeline (default 12) batches are in flight on separate HIP streams
(each step is a complete load + solve of its batch); ms_per_step is total time / steps.
Multi-GPU: independent problems, so each rank owns its own batch (weak scaling, no
data-path collective); RCCL is used only to gather the totals.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NVAR, NI, NPARM = 127, 64, 0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(rows, max_procs):
    """Reference CPU path (oracle/_ref, the real piplib int64 build) on a bounded sample of
    the same workload, one process per host core; falls back to the CPU restatement
    ("port") if the reference build did not travel."""
    import concurrent.futures as cf
    import numpy as np
    import pipbatch as pb
    from piplib_amd import synth
    exe, kind = (pb.REFPIP, "reference") if pb.have_ref() else (pb.ORACLEPIP, "port")
    if not os.access(exe, os.X_OK):
        return None
    cores = max(1, min(max_procs, len(os.sched_getaffinity(0))))
    per = 640  # ~50k pivots per process: ~0.6-1.3 s each, ~10-20 s of CPU work in all, bounded
    n = min(rows.shape[0], per * cores)
    per = max(1, n // cores)
    chunks = [rows[c * per:(c + 1) * per] for c in range(cores)]

    def run(chunk):
        probs = [synth.Problem(NVAR, NPARM, NI, 0, -1, 1, chunk[b], np.zeros((0, NPARM + 1), np.int64))
                 for b in range(chunk.shape[0])]
        return pb.run_batch(exe, probs, pb.F_NOSIMPLIFY | pb.F_NOTEXT)

    t0 = time.time()
    with cf.ThreadPoolExecutor(cores) as ex:
        outs = list(ex.map(run, chunks))
    wall = time.time() - t0
    piv = sum(o.total_pivots for o in outs)
    tmax = max(o.solve_seconds for o in outs)
    return {"value": piv / tmax, "unit": "pivots/s", "cores": cores, "kind": kind,
            "sample": f"first {per * cores} tableaux of rank 0's batch ({piv} pivots), {cores} processes x "
                      f"{per} tableaux, slowest process {tmax:.2f} s solve time (wall {wall:.1f} s incl. I/O)",
            "per_core": piv / sum(o.solve_seconds for o in outs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--batch", type=int, default=10000, help="tableaux per GPU")
    ap.add_argument("--waves", type=int, default=0, help="waves per tableau (0 = engine default)")
    ap.add_argument("--round", type=int, default=0, help="pivot budget per tableau in the bulk launch (0 = engine default)")
    ap.add_argument("--round-rows", type=int, default=0, help="spare rows in the bulk launch's LDS image (0 = engine default)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch tableaux per GPU (default); strong: --batch tableaux in all, sharded over the ranks")
    ap.add_argument("--pipeline", type=int, default=12, help="batches in flight (streams/threads)")
    ap.add_argument("--stagger", type=float, default=-1.0,
                    help="ms between the lanes' starts (0 = none, <0 = step latency / lanes, measured in warm-up)")
    ap.add_argument("--no-dense", action="store_true", help="skip the row-skipping-off measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    # Batches in flight run on separate HIP streams; the runtime multiplexes streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels that share a queue serialise, so
    # give every lane a queue of its own (a HIP runtime setting, read when the runtime starts).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import numpy as np
    import torch
    from piplib_amd import engine as eng
    from piplib_amd import synth

    from piplib_amd import dist as pdist
    rank, world, local = pdist.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # PIPAMD_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than
    # ranks (ranks then share GPUs); the driver's runs use nccl (== RCCL on ROCm), one GPU per rank
    backend = os.environ.get("PIPAMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pdist.init(backend, dev)

    if args.scaling == "strong":  # BASELINE configs[3]: one 10k batch sharded over the GPUs
        lo, hi = pdist.shard_range(args.batch, rank, world)
        rows_h = synth.lexmin_batch(1000, args.batch, NVAR, NI)[lo:hi]
    else:
        rows_h = synth.lexmin_batch(pdist.shard_seed(1000, rank), args.batch, NVAR, NI)
    my_batch = rows_h.shape[0]
    rows_d = torch.as_tensor(rows_h, dtype=torch.int64).to(dev)

    # `depth` batches in flight, each with its own engine, workspace, HIP stream and host thread:
    # while one batch's last stragglers finish (a latency-bound tail that leaves most CUs idle)
    # the next batch's bulk rounds already run.  Every step is still a full load + solve.
    depth = max(1, min(args.pipeline, args.steps))
    # every lane should time the same number of steps: prefer a lane count that divides --steps
    for d in range(depth, max(1, depth // 2) - 1, -1):
        if args.steps % d == 0:
            depth = d
            break
    lanes = []
    for _ in range(depth):
        e = eng.Engine(local)
        if args.waves:
            e.set_waves_per_job(args.waves)
        if args.round:
            e.set_round_pivots(args.round)
        if args.round_rows:
            e.set_round_rows(args.round_rows)
        lanes.append((e, eng.Batch(e, rows_d, NVAR, NPARM, tflags=eng.T_INT), torch.cuda.Stream(dev)))
    e, b, _ = lanes[0]

    def barrier():
        torch.cuda.synchronize(dev)
        pdist.barrier()
        torch.cuda.synchronize(dev)

    stagger = [0.0]

    def run_lane(i, nsteps):
        _, bi, st = lanes[i]
        # lanes start a fraction of a step apart, so that one batch's under-filled last rounds
        # coincide with another batch's bulk rounds instead of with its last rounds
        if stagger[0] > 0 and i:
            time.sleep(i * stagger[0])
        with torch.cuda.stream(st):
            for _ in range(nsteps):
                bi.load()
                bi.solve()
            st.synchronize()

    def run_steps(nsteps):
        import threading
        share = [nsteps // depth + (1 if i < nsteps % depth else 0) for i in range(depth)]
        th = [threading.Thread(target=run_lane, args=(i, share[i])) for i in range(depth) if share[i]]
        for t in th:
            t.start()
        for t in th:
            t.join()

    run_steps(max(args.warmup, depth))
    barrier()
    if depth > 1 and args.stagger != 0:
        if args.stagger > 0:
            stagger[0] = args.stagger * 1e-3
        else:  # one lane's own step latency with every lane busy, spread evenly over the lanes
            tw = time.perf_counter()
            run_steps(depth)
            torch.cuda.synchronize(dev)
            stagger[0] = (time.perf_counter() - tw) / depth
        barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0

    # the advance kernel's own launch durations (HIP events on its stream), un-overlapped
    kernel_ms = []
    for _ in range(2):
        b.load()
        b.solve()
        kernel_ms.append(b.last_solve_ms())
    torch.cuda.synchronize(dev)

    b.fetch()
    torch.cuda.synchronize(dev)
    st = b.status.cpu().numpy()
    piv = int(b.pivots.sum().item())
    cuts = int(b.cuts.sum().item())
    solved = int(((st == eng.ST_SOLUTION) | (st == eng.ST_NIL)).sum())

    rowsw = b.counters()["rows_rewritten"]
    tot, dt_max = pdist.gather_totals([piv, my_batch, solved, cuts, rowsw], dt, dev)

    # Same workload once more with row skipping off: every real row is read and written on
    # every pivot, which is the reference's access pattern (traiter.c:467-502) and the regime
    # in which the row-update path is HBM-bound.  Reported beside the main number.
    dense = None
    if rank == 0 and not args.no_dense:
        ed = eng.Engine(local)
        ed.set_waves_per_job(4)  # streaming regime: four waves share a tableau's rows
        bd = eng.Batch(ed, rows_d, NVAR, NPARM, tflags=eng.T_INT | eng.T_NOSKIP)
        dms = []
        for i in range(3):
            bd.load()
            bd.solve()
            if i:
                dms.append(bd.last_solve_ms())
        cd = bd.counters()
        dk = float(np.mean(dms))
        # every real row (cut rows included, counted by the kernel) is read and written once per
        # pivot, plus the pivot-row read and the write of the row that replaces the unit row
        dbytes = 8.0 * (NVAR + NPARM + 1) * (2.0 * cd["rows_rewritten"] + 2.0 * cd["pivots"])
        dense = {"bound": "hbm", "achieved": dbytes / (dk * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": dbytes / (dk * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": dk, "pivots": cd["pivots"],
                 "rows_rewritten_per_pivot": cd["rows_rewritten"] / max(1, cd["pivots"]),
                 "algorithmic_bytes_per_step": dbytes,
                 "note": "same batch with row skipping disabled (PIPAMD_T_NOSKIP, 4 waves per tableau): every "
                         "real row is read and written on every pivot, the reference's access pattern"}
        del bd, ed

    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        piv_per_step = float(tot[0])
        k_ms = float(np.mean(kernel_ms))
        # Algorithmic HBM bytes of one pivot of THIS algorithm (DESIGN.md "Roofline"): read the
        # pivot row, write the row that replaces the entering unit row, read+write every row that
        # actually changes (counted by the kernel) -- rows with a zero multiplier keep their bits.
        ncol = NVAR + NPARM + 1
        rows_rw = float(tot[4]) / max(1.0, float(world))  # this rank's share is what its launch moved
        piv_rank = piv
        algo_bytes = 8.0 * ncol * (2.0 * b.counters()["rows_rewritten"] + 2.0 * piv_rank)
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")
        if os.path.exists(tp):
            try:
                t = json.load(open(tp))
                if t.get("batch_per_gpu") == my_batch:
                    traffic = t["hbm_bytes_per_step"]
            except Exception:
                traffic = None
        out = {
            "metric": "pivots/sec (batched 64x128 int64 tableaux, integer solve with Gomory cuts)",
            "value": piv_per_step / (ms_step * 1e-3),
            "unit": "pivots/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": "10k-batch synthetic 64x128 tableaux, int64, integer solve with Gomory cuts",
                       "batch_per_gpu": my_batch, "nvar": NVAR, "nparm": NPARM, "ni": NI,
                       "parallelism": f"{world} x independent batches (one workgroup per tableau)",
                       "pipeline_depth": depth},
            "problems_per_sec": float(tot[1]) / (ms_step * 1e-3),
            "pivots_per_step": piv_per_step,
            "cuts_per_step": float(tot[3]),
            "finished_fraction": float(tot[2]) / float(tot[1]),
            "rows_rewritten_per_pivot": float(tot[4]) / max(1.0, piv_per_step),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "pip_advance_kernel", "kernel_ms": k_ms,
                         "launches_per_step": e.last_solve_launches(),
                         "avg_launch_ms": k_ms / max(1, e.last_solve_launches()),
                         "measured": "HIP events around each launch, 2 un-pipelined steps after the timed "
                                     "region (= `bench.py --pipeline 1`, the command of profiles/r01_kernel_stats.csv)",
                         "algorithmic_bytes_per_step": algo_bytes,
                         "timed_region_GBps": algo_bytes / (ms_step * 1e-3) / 1e9,
                         "dense_equivalent_GBps": b.pivot_bytes() * piv_rank / (k_ms * 1e-3) / 1e9,
                         "note": "sparse workload: ~2.7 of ~80 rows change per pivot, so the pivot loop is "
                                 "latency/issue-bound, not HBM-bound; see roofline_dense_mode for the "
                                 "HBM-bound regime of the same kernel"},
        }
        if dense:
            out["roofline_dense_mode"] = dense
        if not args.no_cpu:
            cb = cpu_baseline(rows_h, 16)
            if cb:
                out["cpu_baseline"] = cb
                out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        print(json.dumps(out), flush=True)
    pdist.finish()


if __name__ == "__main__":
    main()
