"""Manual GPU fuzz (not a test): random shapes / densities / magnitudes, HIP engine vs the CPU
oracle, bit-exact (status, pivots, solution); plus skipping off and 128-bit entries against the
int64 engine.  Usage: python tests/manual/fuzz_gpu.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from piplib_amd import engine as eng, synth
import gpu_common as gc
import pipbatch as pb

import subprocess


def _oracle_batch(rows, nvar, nparm, nq, bigparm=-1):  # bounded: some dense integer problems cut (almost) forever
    probs = [synth.Problem(nvar, nparm, rows.shape[1], 0, bigparm, nq, rows[k], np.zeros((0, nparm + 1), np.int64))
             for k in range(rows.shape[0])]
    return pb.run_batch(pb.ORACLEPIP, probs, pb.F_NOSIMPLIFY, timeout=15)


gc.oracle_batch = _oracle_batch
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); ncase = nprob = npiv = 0
e = eng.Engine(0)
while time.time() - t0 < budget:
    nvar = int(rng.choice([3, 5, 9, 17, 30, 63, 64, 65, 100, 127, 128, 129, 140]))
    ni = int(rng.integers(2, 100))
    nq = int(rng.integers(0, 2))
    dense = rng.random() < 0.25
    seed = int(rng.integers(1, 1 << 30))
    batch = 24
    if dense:
        nvar, ni = min(nvar, 30), min(ni, 24)
        rows = synth.dense_batch(seed, batch, nvar, ni, cmax=int(rng.choice([3, 10, 40])), x0max=9, pzero=float(rng.choice([0.3, 0.6])))
        kw = "dense"
    else:
        kw = dict(nnz=int(rng.integers(2, 7)), cmax=int(rng.choice([1, 2, 5, 30])), x0max=int(rng.choice([3, 9, 50])),
                  slackmax=int(rng.choice([0, 3, 10])), pneg=float(rng.choice([0.0, 0.25, 0.5])))
        rows = synth.lexmin_batch(seed, batch, nvar, ni, **kw)
    tag = f"nvar={nvar} ni={ni} nq={nq} seed={seed} {kw}"
    try:
        n, piv = gc.compare(rows, nvar, 0, nq, cap_cuts=600)
    except subprocess.TimeoutExpired:
        print("skipped (oracle needs more than 15 s):", tag, flush=True)
        continue
    except AssertionError as ex:
        a = ex.args[0] if ex.args else ()
        if isinstance(a, tuple) and len(a) >= 2 and int(a[1]) == eng.ST_CAPACITY:
            print("skipped (a tableau needs more than 600 cut rows: PIPAMD_ST_CAPACITY):", tag, flush=True)
            continue
        print("MISMATCH vs oracle:", tag, repr(ex)[:300], flush=True)
        sys.exit(1)
    except Exception as ex:
        print("MISMATCH vs oracle:", tag, repr(ex)[:300], flush=True)
        sys.exit(1)
    # same batch with skipping off against the int64 engine ...
    ref = eng.Batch(e, rows, nvar, 0, tflags=eng.T_INT if nq else 0, cap_cuts=600); ref.load(); ref.solve(); ref.fetch()
    b = eng.Batch(e, rows, nvar, 0, tflags=(eng.T_INT if nq else 0) | eng.T_NOSKIP | eng.T_ROWS_STAY, cap_cuts=600); b.load(); b.solve(); b.fetch()  # (rows fetched by the pivot kernel where the shape allows)
    torch.cuda.synchronize()
    st, st0 = b.status.cpu().numpy(), ref.status.cpu().numpy()
    fin = st0 == eng.ST_SOLUTION
    if not ((st == st0).all() and (b.pivots.cpu().numpy() == ref.pivots.cpu().numpy()).all()
            and (b.sol_num.cpu().numpy()[fin] == ref.sol_num.cpu().numpy()[fin]).all()
            and (b.sol_den.cpu().numpy()[fin] == ref.sol_den.cpu().numpy()[fin]).all()):
        print("MISMATCH noskip vs skipping engine:", tag, flush=True)
        sys.exit(1)
    # ... and on 128-bit entries against the 128-bit oracle
    probs = [synth.Problem(nvar, 0, ni, 0, -1, nq, rows[k], np.zeros((0, 1), np.int64)) for k in range(rows.shape[0])]
    try:
        o128 = pb.run_batch(pb.ORACLEPIP128, probs, pb.F_NOSIMPLIFY, timeout=15)
    except subprocess.TimeoutExpired:
        print("skipped the 128-bit leg (oracle needs more than 15 s):", tag, flush=True)
        continue
    g = eng.Batch(e, rows, nvar, 0, tflags=eng.T_INT if nq else 0, entier_bits=128, cap_cuts=600); g.load(); g.solve(); g.fetch()
    torch.cuda.synchronize()
    st, pv = g.status.cpu().numpy(), g.pivots.cpu().numpy()
    num, den = eng.wide_to_int(g.sol_num.cpu().numpy()), eng.wide_to_int(g.sol_den.cpu().numpy())
    for k, r in enumerate(o128.results):
        if st[k] == eng.ST_CAPACITY:
            continue
        if r.status == pb.ST_ABORT:
            bad = st[k] != {2: eng.ST_OVERFLOW, 4: eng.ST_MAXCOL}.get(r.abort_code, eng.ST_OVERFLOW)
        else:
            got = "()" if st[k] == eng.ST_NIL else pb.squash(gc.solution_text(num[k], den[k]))
            bad = pv[k] != r.pivots or got != pb.squash(r.text)
        if bad:
            print(f"MISMATCH int128 engine vs 128-bit oracle: tableau {k}", tag, flush=True)
            sys.exit(1)
    ncase += 1; nprob += n; npiv += piv
    if ncase % 10 == 0:
        print(f"{ncase} cases, {nprob} tableaux, {npiv} pivots, {time.time()-t0:.0f} s", flush=True)
print(f"OK: {ncase} cases, {nprob} tableaux, {npiv} pivots checked")
