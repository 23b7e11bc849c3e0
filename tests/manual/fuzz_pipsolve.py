"""Manual GPU fuzz (not a test): the reference's pip_solve with traiter() bound to the GPU engine
(oracle/_ref/refpip_gpu, bindings/piplib_traiter_hook.c) with random option sets (Maximize /
Urs_unknowns / Urs_parms / Rational / Dual / bignum) against the CPU oracle's `pip` mode, which
prints what the reference's example.c prints.
Usage: python tests/manual/fuzz_pipsolve.py [seconds] [seed] [big]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from piplib_amd import engine as eng
import pipbatch as pb
from datfile import matrix_text

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BIG = len(sys.argv) > 3  # larger shapes and coefficients: more aborts and skips, deeper trees
t0 = time.time(); n = nskip = nabort = nsplit = 0
while time.time() - t0 < budget:
    nn, npar = int(rng.integers(1, 9 if BIG else 6)), int(rng.integers(0, 5 if BIG else 4))
    nrow, ncrow = int(rng.integers(1, 13 if BIG else 9)), int(rng.integers(0, 4)) if npar else 0
    cm = int(rng.choice([3, 6, 12])) if BIG else 3
    dom = rng.integers(-cm, cm + 1, size=(nrow, nn + npar + 2)).astype(np.int64)
    dom[:, 0] = rng.random(nrow) < 0.85
    dom[:, -1] = rng.integers(-6, 12, size=nrow)
    u = rng.random()
    if u < 0.06:    # degenerate inputs: an all-zero row, a duplicated row, no rows at all
        dom[int(rng.integers(0, nrow)), 1:] = 0
    elif u < 0.12 and nrow > 1:
        dom[int(rng.integers(0, nrow))] = dom[int(rng.integers(0, nrow))]
    elif u < 0.16:
        dom = dom[:0]
    ctx = rng.integers(-2, 3, size=(ncrow, npar + 2)).astype(np.int64)
    if ncrow:
        ctx[:, 0] = 1
        ctx[:, -1] = rng.integers(0, 9, size=ncrow)
    opts = {}
    words = ""
    if rng.random() < 0.25: opts["Maximize"] = 1; words += "Maximize\n"
    if rng.random() < 0.25: opts["Urs_unknowns"] = 1; words += "Urs_unknowns\n"
    if npar and rng.random() < 0.25: opts["Urs_parms"] = 1; words += "Urs_parms\n"
    if rng.random() < 0.3:
        opts["Nq"] = 0; words += "Rational\n"
        # (not with Urs_parms: the reference itself exits or faults on that combination)
        if "Urs_parms" not in opts and rng.random() < 0.4: opts["Compute_dual"] = 1; words += "Dual\n"
    # the bignum column is given in context-matrix columns (example.c); pip_solve wants it in
    # domain-matrix columns
    bignum = int(rng.integers(1, npar + 1)) if (npar and rng.random() < 0.2) else -1
    bg = bignum + (dom.shape[1] - ctx.shape[1]) if bignum > 0 else bignum
    txt = (matrix_text(ctx) + f"\n{bignum}\n\n" + matrix_text(dom) + "\n" + words).encode()
    try:
        o = subprocess.run([pb.ORACLEPIP, "pip"], input=txt, capture_output=True, timeout=3)
    except subprocess.TimeoutExpired:
        nskip += 1
        continue
    tag = f"dom={dom.tolist()} ctx={ctx.tolist()} opts={opts} bignum={bignum}"
    if os.environ.get("FUZZ_LAST_CASE"):  # a crash in the library leaves the input behind
        with open(os.environ["FUZZ_LAST_CASE"], "w") as f:
            f.write(tag + "\n")
    g = subprocess.run([pb.REFPIP_GPU, "pip"], input=txt, capture_output=True, timeout=600)
    if g.returncode != 0:
        if o.returncode == 0 and b"status 6" in g.stderr:
            print("skipped (PIPAMD_ST_CAPACITY: more than 16,000 rows):", tag, flush=True)
            nskip += 1
            continue
        if o.returncode == 0:
            print("MISMATCH: engine aborted, oracle did not:", tag, g.stderr.decode()[-200:], flush=True); sys.exit(1)
        nabort += 1
        continue
    if o.returncode != 0:
        print("MISMATCH: oracle aborted, engine did not:", tag, flush=True); sys.exit(1)
    text = g.stdout.decode("latin-1")
    want = o.stdout.decode("latin-1")
    if pb.squash(text) != pb.squash(want):
        print("MISMATCH:", tag, "\n got ", text[-400:], "\n want", want[-400:], flush=True); sys.exit(1)
    n += 1; nsplit += "(if" in text
    if n % 200 == 0:
        print(f"{n} problems ({nsplit} with splits, {nabort} aborts, {nskip} skipped), {time.time()-t0:.0f} s", flush=True)
print(f"OK: {n} problems ({nsplit} with splits, {nabort} aborts on both sides, {nskip} skipped) checked")
