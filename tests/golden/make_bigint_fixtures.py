#!/usr/bin/env python3
"""Generate tests/golden/bigint/*.json: exact-arithmetic results that pin the 128-bit Entier engine.

    python tests/golden/make_bigint_fixtures.py [--check-only] [family ...]

The checker is tests/bigint_pip.py (Python ints, the reference's pivot rule restated from
source/traiter.c and source/integrer.c).  Before anything is written it is validated here, in the
build container, against

  * oracle/oraclepip (the C restatement pinned to the reference's golden .ll files) on inputs that
    stay inside 64 bits: status, pivot count and solution text must agree, and
  * the reference itself (oracle/_ref/refpip) on the same inputs when it is built.

Fixture families (inputs are regenerated from the seed by piplib_amd/synth.py, so only the
expected outputs are stored):

  dense10   synth.dense_batch(seed, 48, 10, 10)   -- determinants outgrow 64 bits, the int64 build
  dense14   synth.dense_batch(seed, 24, 12, 14)      stops with "Integer overflow" on many of them
  wide128   synth.lexmin_batch(seed, 1040, 255, 128, nnz=16, cmax=30), screened (see FAMILIES)
            -- BASELINE configs[4]'s shape (128x256) with coefficients that drive tableau entries
               beyond 2^63 in many tableaux (`entry_bits` per tableau is recorded)

Per tableau: status (PIPAMD_ST_*), pivots, cuts, the solution as decimal strings, the largest
entry and the largest intermediate (bits), and `exact` = no intermediate reached 128 bits, i.e. a
128-bit fixed-width run cannot have wrapped and must reproduce the record bit for bit.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import bigint_pip as bp  # noqa: E402
import pipbatch as pb  # noqa: E402
from piplib_amd import synth  # noqa: E402

FAMILIES = {
    "dense10": dict(gen="dense_batch", seed=101, batch=48, nvar=10, ni=10, kw={}),
    "dense14": dict(gen="dense_batch", seed=202, batch=24, nvar=12, ni=14, kw={}),
    # BASELINE configs[4]'s shape.  Gomory cuts converge slowly on a few tableaux of any such family
    # (tens of thousands of pivots): 1,040 candidates are drawn, those the 128-bit C oracle does not
    # finish within SCREEN_SECONDS are listed in the fixture ("skip") and left out, and the first 1,000
    # of the rest are the batch tests/test_gpu_parity.py solves at full size; the first 24 of those
    # carry exact-arithmetic records.
    "wide128": dict(gen="lexmin_batch", seed=303, batch=1040, screen=1000, take=24, nvar=255, ni=128,
                    kw=dict(nnz=16, cmax=30)),
}
SCREEN_SECONDS = 8


def _skip_list(fam):
    path = os.path.join(HERE, "bigint", fam + ".json")
    if os.path.exists(path):
        return json.load(open(path)).get("skip", [])
    return None


def rows_full(fam, skip=None):
    """the whole batch of a family (screened families: candidates minus the fixture's skip list)"""
    f = FAMILIES[fam]
    rows = getattr(synth, f["gen"])(f["seed"], f["batch"], f["nvar"], f["ni"], **f["kw"])
    if "screen" in f:
        skip = _skip_list(fam) if skip is None else skip
        assert skip is not None, "fixture with the skip list not generated yet"
        keep = [b for b in range(rows.shape[0]) if b not in set(skip)][:f["screen"]]
        rows = rows[keep]
    return rows


def rows_of(fam, skip=None):
    """the tableaux of a family that carry exact-arithmetic records"""
    rows = rows_full(fam, skip)
    return rows[:FAMILIES[fam].get("take", rows.shape[0])]


def screen(fam):
    """indices of the candidates the 128-bit C oracle does not finish within SCREEN_SECONDS"""
    import concurrent.futures as cf
    import subprocess
    f = FAMILIES[fam]
    rows = getattr(synth, f["gen"])(f["seed"], f["batch"], f["nvar"], f["ni"], **f["kw"])

    def slow(b):
        p = [synth.Problem(f["nvar"], 0, f["ni"], 0, -1, 1, rows[b], np.zeros((0, 1), np.int64))]
        try:
            pb.run_batch(pb.ORACLEPIP128, p, pb.F_NOSIMPLIFY | pb.F_NOTEXT, timeout=SCREEN_SECONDS)
            return False
        except subprocess.TimeoutExpired:
            return True
    with cf.ThreadPoolExecutor(max(1, len(os.sched_getaffinity(0)))) as ex:
        flags = list(ex.map(slow, range(rows.shape[0])))
    return [b for b, s_ in enumerate(flags) if s_]


def validate_against_oracles():
    """bits=64 runs of the Python checker vs the pinned C oracle and the real reference."""
    cases = [("lexmin 31x16 integer", synth.lexmin_batch(7, 40, 31, 16), 1),
             ("lexmin 31x16 rational", synth.lexmin_batch(8, 40, 31, 16), 0),
             ("dense 10x10", synth.dense_batch(3, 40, 10, 10), 1),
             ("dense 12x14", synth.dense_batch(4, 20, 12, 14), 1),
             ("bench 64x128", synth.lexmin_batch(1000, 4, 127, 64), 1)]
    exes = [pb.ORACLEPIP] + ([pb.REFPIP] if pb.have_ref() else [])
    for tag, rows, nq in cases:
        nvar = rows.shape[2] - 1
        probs = [synth.Problem(nvar, 0, rows.shape[1], 0, -1, nq, rows[b], np.zeros((0, 1), np.int64))
                 for b in range(rows.shape[0])]
        mine = [bp.solve(rows[b], integer=bool(nq), bits=64) for b in range(rows.shape[0])]
        for exe in exes:
            o = pb.run_batch(exe, probs, pb.F_NOSIMPLIFY)
            n = 0
            for b, r in enumerate(o.results):
                m = mine[b]
                if r.status == pb.ST_ABORT:
                    assert m.status == bp.ST_OVERFLOW or not m.stats.exact, (tag, b)
                    continue
                if not m.stats.exact:
                    continue  # the 64-bit run wrapped somewhere: not comparable
                assert m.pivots == r.pivots, (tag, b, m.pivots, r.pivots)
                want = pb.squash(r.text)
                if want == "()":
                    assert m.status == bp.ST_NIL, (tag, b)
                else:
                    assert m.status == bp.ST_SOLUTION, (tag, b, m.status)
                    assert pb.squash(bp.solution_text(m.sol_num, m.sol_den)) == want, (tag, b)
                n += 1
            print(f"validated {tag}: {n}/{len(o.results)} comparable tableaux agree with {os.path.basename(exe)}")


def make(fam):
    skip = screen(fam) if "screen" in FAMILIES[fam] else None
    rows = rows_of(fam, skip)
    out = []
    t0 = time.time()
    for b in range(rows.shape[0]):
        r = bp.solve(rows[b], integer=True, bits=128)
        out.append({"status": r.status, "pivots": r.pivots, "cuts": r.cuts,
                    "sol_num": [str(x) for x in r.sol_num] if r.sol_num is not None else None,
                    "sol_den": [str(x) for x in r.sol_den] if r.sol_den is not None else None,
                    "entry_bits": r.stats.max_entry_bits, "max_bits": r.stats.max_bits, "exact": r.stats.exact})
    f = FAMILIES[fam]
    doc = {"family": fam, "generator": f"synth.{f['gen']}({f['seed']}, {f['batch']}, {f['nvar']}, {f['ni']}"
                                       + "".join(f", {k}={v}" for k, v in f["kw"].items()) + ")"
                                       + (f" minus `skip`, first {f['screen']}; records for the first {f['take']}"
                                          if "screen" in f else ""),
           "bits": 128, "made_by": "tests/golden/make_bigint_fixtures.py (tests/bigint_pip.py, Python ints)",
           "tableaux": out}
    if skip is not None:
        doc["skip"] = skip
        doc["skip_reason"] = f"oraclepip128 did not finish the tableau within {SCREEN_SECONDS} s when the fixture was made"
    print(f"{fam}: {len(out)} tableaux in {time.time() - t0:.0f} s; "
          f"{sum(t['exact'] for t in out)} exact at 128 bits, "
          f"{sum(t['entry_bits'] > 63 for t in out)} with entries beyond 2^63, "
          f"statuses {sorted(set(t['status'] for t in out))}")
    return doc


def main():
    validate_against_oracles()
    if "--check-only" in sys.argv:
        return
    os.makedirs(os.path.join(HERE, "bigint"), exist_ok=True)
    for fam in ([a for a in sys.argv[1:] if a in FAMILIES] or FAMILIES):
        doc = make(fam)
        with open(os.path.join(HERE, "bigint", fam + ".json"), "w") as f:
            json.dump(doc, f, indent=0)
            f.write("\n")


if __name__ == "__main__":
    main()
