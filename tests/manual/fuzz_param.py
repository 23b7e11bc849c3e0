"""Manual GPU fuzz (not a test): random parametric problems through the device-resident traiter()
(csrc/pip_quast.hip, in front of both many-problem entries) and through the host decision tree alone
(lock-step forest and per-problem trees) against the CPU oracle: same sol_edit text, same pivot
count, same abort verdict.  Usage: python tests/manual/fuzz_param.py [seconds] [seed]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from piplib_amd import engine as eng, synth
import pipbatch as pb

BITS = int(os.environ.get("BITS", "64"))  # 128: the overflow-safe flavour (128-bit oracle, *_lockstep128 / *_tableaux128, the 128-bit device tree)
ORACLE = pb.ORACLEPIP128 if BITS == 128 else pb.ORACLEPIP
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
e = eng.Engine(0)
t0 = time.time(); ncase = nprob = npiv = nsplit = nserved = nback = 0
while time.time() - t0 < budget:
    nvar = int(rng.integers(2, 12)); nparm = int(rng.integers(0, 5)); ni = int(rng.integers(2, 14))
    if os.environ.get("FUZZ_BIG"):  # larger tableaux: more than 64 logical rows, up to 50 columns
        nvar = int(rng.integers(8, 40)); nparm = int(rng.integers(0, 7)); ni = int(rng.integers(6, 50))
    if os.environ.get("FUZZ_TALL"):  # 57 ... 104 inequalities: the device tree's two-rows-per-lane sort (sort_rows_tall)
        nvar = int(rng.integers(3, 24)); nparm = int(rng.integers(0, 5)); ni = int(rng.integers(57, 105))
    nc = int(rng.integers(0, 4)) if nparm else 0
    nq = int(rng.integers(0, 2)); deepest = bool(nq and rng.random() < 0.25)
    seed = int(rng.integers(1, 1 << 30))
    cmax = int(rng.choice([2, 4, 9])); bmax = int(rng.choice([5, 12, 40]))
    if os.environ.get("FUZZ_BIG"):
        cmax = int(rng.choice([1, 1, 2]))
    tag = f"nvar={nvar} nparm={nparm} ni={ni} nc={nc} nq={nq} deepest={deepest} seed={seed} cmax={cmax} bmax={bmax}"
    probs, want = [], []
    for p in synth.random_problems(seed, 8 if (os.environ.get("FUZZ_BIG") or os.environ.get("FUZZ_TALL")) else 24, nvar, nparm, ni, nc, nq, cmax=cmax, bmax=bmax):
        try:  # some random parametric problems make the reference itself cut forever
            r = pb.run_batch(ORACLE, [p], pb.F_DEEPEST if deepest else 0, timeout=2).results[0]
        except subprocess.TimeoutExpired:
            continue
        if r.pivots <= 2000:
            probs.append(p); want.append(r)
    if not probs:
        continue
    if nparm and rng.random() < 0.3:  # a big parameter (any parameter column)
        bp = nvar + 1 + int(rng.integers(0, nparm))
        probs = [pb.Problem(p.nvar, p.nparm, p.ni, p.nc, bp, p.nq, p.ineq, p.ctx) for p in probs]
        try:
            want = [pb.run_batch(ORACLE, [p], pb.F_DEEPEST if deepest else 0, timeout=4).results[0] for p in probs]
        except subprocess.TimeoutExpired:
            continue
        tag += f" bigparm={bp}"
    for mode, dt in (("lockstep", True), ("threads", True), ("lockstep", False), ("threads", False)):
        e.set_device_tree(dt)
        if os.environ.get("FUZZ_VERBOSE"):
            print(f"[{time.time()-t0:.1f} s] {mode} device tree {dt}: {len(probs)} problems, oracle pivots {[r.pivots for r in want]}", tag, flush=True)
        if BITS == 128:
            got = (eng.solve_tableaux_lockstep128(e, probs, simplify=True, deepest_cut=deepest) if mode == "lockstep" else
                   eng.solve_tableaux128(e, probs, simplify=True, deepest_cut=deepest, nthreads=4))
        else:
            got = eng.solve_tableaux(e, probs, simplify=True, deepest_cut=deepest, lockstep=(mode == "lockstep"), nthreads=4)
        if dt and mode == "lockstep":
            sv, bk = e.last_device_tree()
            nserved += sv; nback += bk
        for i, ((text, rc, st, piv), r) in enumerate(zip(got, want)):
            if r.status == pb.ST_ABORT:
                bad = rc == 0
            else:
                w = "void" if r.status == pb.ST_VOID else pb.squash(r.text)
                bad = rc != 0 or pb.squash(text or "") != w or piv != r.pivots
            if bad:
                print(f"MISMATCH ({mode}, device tree {dt}) problem {i}:", tag, "rc", rc, "status", st, "pivots", piv, r.pivots, flush=True)
                print(" got ", (text or "")[:300]); print(" want", r.text[:300])
                sys.exit(1)
    ncase += 1; nprob += len(probs); npiv += sum(r.pivots for r in want); nsplit += sum("if" in r.text for r in want)
    if ncase % 10 == 0:
        print(f"{ncase} cases, {nprob} problems ({nsplit} with splits), {npiv} pivots, device tree served {nserved} / handed back {nback}, {time.time()-t0:.0f} s", flush=True)
print(f"OK: {ncase} cases, {nprob} problems ({nsplit} with splits), {npiv} pivots checked; device tree served {nserved}, handed back {nback}")
