#!/usr/bin/env python3
"""Generate tests/golden/bench_screen.json: which tableaux of bench.py's headline batches are left out of the
timed workload, decided by the CPU oracle (oracle/oraclepip, the restatement pinned to the reference) and by nothing
of the engine under test.

    python tests/golden/make_bench_screen.py [first_batch [last_batch]]

bench.py's batch g (BASELINE configs[2]'s shape: 10,000 tableaux of 64 rows x 127 unknowns + constant) is
synth.lexmin_batch(1000 + 7919 * g, ...).  On about 3 tableaux in 100,000 of that family Gomory's cuts do not
converge: the reference's own loop (traiter.c:665-789 with integrer.c:410-415's unbounded expanser) does not finish
them within minutes.  A benchmark workload must be one the reference finishes, so those tableaux are replaced by
their neighbours.  The criterion is a property of the algorithm, not of an implementation: a tableau is "slow" when
integrer() asks for a 449th constant cut (every correct implementation adds the same cuts in the same order).  The
oracle runs every tableau with ORACLE_MAX_CUTS=448 and reports the ones that stop on that budget; per batch the file
keeps

    slow             indices of those tableaux
    pivots_finishing pivots (calls of pivoter) the oracle needs for all the others
    pivots_screened  pivots of the batch bench.py times: every slow tableau replaced by the next tableau that is not

bench.py checks the engine against all three: the set of tableaux it leaves at PIPAMD_ST_CAPACITY under the same row
budget must be `slow`, and the pivots of its pre-pass and of its timed regions must add up to `pivots_screened`.
"""
import concurrent.futures as cf
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import pipbatch as pb  # noqa: E402
from piplib_amd import synth  # noqa: E402

BATCH, NVAR, NI, MAX_CUTS = 10000, 127, 64, 448
NBATCH = 128   # weak scaling: 8 ranks x 16 lanes; strong scaling: lanes x fused batches
OUT = os.path.join(HERE, "bench_screen.json")
WIDE128_MAX_CUTS = 1024  # configs[4]'s pinned batch: every tableau finishes within that many cuts (9 need more than 448)
ORA_ERR_BUDGET = 9


def seed_of(g):
    return 1000 + 7919 * g


def replacement(b, slow, n):
    """bench.py's rule: the next tableau (cyclically) that is not slow itself"""
    return next(x % n for x in range(b + 1, b + n) if x % n not in slow)


def one_batch(g, workers):
    rows = synth.lexmin_batch(seed_of(g), BATCH, NVAR, NI)
    per = (BATCH + workers - 1) // workers

    def run(c):
        probs = [synth.Problem(NVAR, 0, NI, 0, -1, 1, rows[b], np.zeros((0, 1), np.int64))
                 for b in range(c * per, min(BATCH, (c + 1) * per))]
        return pb.run_batch(pb.ORACLEPIP, probs, pb.F_NOSIMPLIFY | pb.F_NOTEXT, timeout=3600).results
    with cf.ThreadPoolExecutor(workers) as ex:
        res = [r for part in ex.map(run, range(workers)) for r in part]
    assert len(res) == BATCH
    slow = [b for b, r in enumerate(res) if r.status == pb.ST_ABORT and r.abort_code == ORA_ERR_BUDGET]
    other = [b for b, r in enumerate(res) if r.status != pb.ST_OK and b not in slow]
    assert not other, ("oracle neither finished nor ran out of cuts", g, other[:5])
    piv = [r.pivots for r in res]
    fin = sum(p for b, p in enumerate(piv) if b not in slow)
    scr = fin + sum(piv[replacement(b, set(slow), BATCH)] for b in slow)
    return {"seed": seed_of(g), "slow": slow, "pivots_finishing": fin, "pivots_screened": scr}


def wide128(workers):
    """BASELINE configs[4]'s pinned batch (tests/golden/make_bigint_fixtures.py rows_full("wide128"): the 1,000 tableaux
    tests/golden/gmp/wide128.json holds the reference's GMP outputs for) through the 128-bit oracle under a budget of
    WIDE128_MAX_CUTS cuts (nothing is left out: `slow` is empty): `unfinished` = tableaux that end any other way than solution / nil (a determinant beyond three 128-bit limbs,
    traiter.c:412-446), `pivots_screened` = pivots of all the others"""
    sys.path.insert(0, HERE)
    import make_bigint_fixtures as mk
    rows = mk.rows_full("wide128")
    n, nvar = rows.shape[0], rows.shape[2] - 1
    per = (n + workers - 1) // workers

    def run(c):
        probs = [synth.Problem(nvar, 0, rows.shape[1], 0, -1, 1, rows[b], np.zeros((0, 1), np.int64))
                 for b in range(c * per, min(n, (c + 1) * per))]
        return pb.run_batch(pb.ORACLEPIP128, probs, pb.F_NOSIMPLIFY | pb.F_NOTEXT, timeout=7200).results
    with cf.ThreadPoolExecutor(workers) as ex:
        res = [r for part in ex.map(run, range(workers)) for r in part]
    slow = [b for b, r in enumerate(res) if r.status == pb.ST_ABORT and r.abort_code == ORA_ERR_BUDGET]
    sl = set(slow)
    rep = {b: replacement(b, sl, n) for b in slow}
    eff = [res[rep.get(b, b)] for b in range(n)]
    unfinished = [b for b, r in enumerate(eff) if r.status != pb.ST_OK]
    return {"slow": slow, "unfinished": unfinished, "unfinished_abort_codes": sorted({eff[b].abort_code for b in unfinished}),
            "pivots_screened": sum(r.pivots for r in eff if r.status == pb.ST_OK)}


def main():
    os.environ["ORACLE_MAX_CUTS"] = str(MAX_CUTS)
    if len(sys.argv) > 1 and sys.argv[1] == "wide128":
        os.environ["ORACLE_MAX_CUTS"] = str(WIDE128_MAX_CUTS)
        doc = json.load(open(OUT))
        t0 = time.time()
        doc["wide128"] = {"0": dict(wide128(max(1, len(os.sched_getaffinity(0)))), max_cuts=WIDE128_MAX_CUTS)}
        print("wide128:", {k: (v if not isinstance(v, list) or len(v) < 20 else len(v)) for k, v in doc["wide128"]["0"].items()},
              f"{time.time() - t0:.0f} s")
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=0, sort_keys=True)
            f.write("\n")
        return
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else NBATCH
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {
        "made_by": "tests/golden/make_bench_screen.py: oracle/oraclepip with ORACLE_MAX_CUTS=%d on "
                   "synth.lexmin_batch(1000 + 7919 * g, %d, %d, %d)" % (MAX_CUTS, BATCH, NVAR, NI),
        "max_cuts": MAX_CUTS, "batch": BATCH, "nvar": NVAR, "ni": NI, "batches": {}}
    workers = max(1, len(os.sched_getaffinity(0)))
    for g in range(lo, hi):
        t0 = time.time()
        doc["batches"][str(g)] = rec = one_batch(g, workers)
        print(f"batch {g} (seed {rec['seed']}): slow {rec['slow']}, {rec['pivots_screened']} pivots, {time.time() - t0:.0f} s",
              flush=True)
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=0, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
