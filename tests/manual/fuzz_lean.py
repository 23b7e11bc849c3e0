"""Manual GPU fuzz (not a test) of the lean bulk kernel (csrc/pip_lean.h): random shapes up to 127 unknowns, densities and
magnitudes around the kernel's limits (entries near 2^15, 2^31, beyond 32 bits; rows multiplied through by powers of two),
every batch through pipamd_batch_solve's bulk launch sequence with and without the lean kernel (identical statuses, pivot
and cut counts, solutions) and against the CPU oracle.  Usage: python tests/manual/fuzz_lean.py [seconds] [seed]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from piplib_amd import engine as eng, synth
from gpu_common import solution_text
import pipbatch as pb

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); ncase = nprob = npiv = nlean = 0
while time.time() - t0 < budget:
    nvar = int(rng.choice([1, 2, 5, 9, 17, 30, 41, 62, 63, 64, 100, 126, 127]))
    ni = int(rng.integers(1, 113))
    nq = int(rng.integers(0, 2))
    seed = int(rng.integers(1, 1 << 30))
    batch = 96
    kw = dict(nnz=int(rng.integers(2, 7)), cmax=int(rng.choice([1, 2, 5, 30, 200, 5000, 30000])), x0max=int(rng.choice([3, 9, 50])),
              slackmax=int(rng.choice([0, 3, 10])), pneg=float(rng.choice([0.0, 0.25, 0.5])))
    rows = synth.lexmin_batch(seed, batch, nvar, ni, **kw)
    scale = int(rng.choice([1, 1, 1 << 7, 1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 20, 1 << 24, 1 << 27, 1 << 30, 1 << 33]))
    if scale > 1:  # some inequalities multiplied through (same polyhedron, larger entries)
        which = rng.random((batch, ni)) < 0.15
        rows = np.where(which[:, :, None], rows * scale, rows)
    stay = bool(rng.integers(0, 2))
    cap = int(rng.choice([4, 40, 600]))
    tag = f"nvar={nvar} ni={ni} nq={nq} seed={seed} scale={scale} stay={stay} cap={cap} {kw}"
    probs = [synth.Problem(nvar, 0, rows.shape[1], 0, -1, nq, rows[k], np.zeros((0, 1), np.int64)) for k in range(batch)]
    try:
        o = pb.run_batch(pb.ORACLEPIP, probs, pb.F_NOSIMPLIFY, timeout=15).results
    except subprocess.TimeoutExpired:
        print("skipped (oracle needs more than 15 s):", tag, flush=True)
        continue
    outs, launches = [], []
    for lean in (0, 1):
        e = eng.Engine(0)
        e.set_bulk_min(64)
        e.set_max_rows(ni + 2048)
        e.debug_lean(lean)
        b = eng.Batch(e, rows, nvar, 0, tflags=(eng.T_INT if nq else 0) | (eng.T_ROWS_STAY if stay else 0), cap_cuts=cap)
        b.load(); b.solve(); launches.append(e.last_solve_launches()); b.fetch()
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (b.status, b.pivots, b.cuts, b.sol_num, b.sol_den)])
    nlean += launches[1] > launches[0]
    if not all((x == y).all() for x, y in zip(*outs)):
        print("MISMATCH lean vs general kernel:", tag, flush=True)
        sys.exit(1)
    st, pv, _, num, den = outs[1]
    for k, r in enumerate(o):
        if st[k] == eng.ST_CAPACITY:
            continue  # the row budget of this run (the oracle has none: it may go on to a solution or to "Integer overflow")
        if r.status == pb.ST_ABORT:
            want = {2: eng.ST_OVERFLOW, 4: eng.ST_MAXCOL}.get(r.abort_code, eng.ST_OVERFLOW)
            ok = st[k] == want
        else:
            ok = st[k] in (eng.ST_SOLUTION, eng.ST_NIL) and pv[k] == r.pivots and \
                ("()" if st[k] == eng.ST_NIL else pb.squash(solution_text(num[k], den[k]))) == pb.squash(r.text)
        if not ok:
            print("MISMATCH vs oracle:", tag, "tableau", k, "status", st[k], "pivots", pv[k], "oracle", r.status, r.pivots, flush=True)
            sys.exit(1)
    ncase += 1; nprob += batch; npiv += int(pv.sum())
    if ncase % 20 == 0:
        print(f"{ncase} cases, {nprob} tableaux, {npiv} pivots, {nlean} cases with a lean launch, {time.time() - t0:.0f} s", flush=True)
print(f"done: {ncase} cases, {nprob} tableaux, {npiv} pivots, {nlean} cases with a lean launch, all bit-exact")
