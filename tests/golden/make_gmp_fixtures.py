#!/usr/bin/env python3
"""Generate tests/golden/gmp/*.json: outputs of the REFERENCE's own arbitrary-precision build
(-DPIPLIB_INT_GMP, include/piplib/piplib.h:40-88; oracle/_ref/refpip_gmp = oracle/ref_driver.c over the
five reference sources where they lie, `make -C oracle ref_gmp`) on the inputs that exercise the
128-bit Entier engine.  Build container only: the GPU box has neither /root/reference nor (necessarily)
libgmp, so the outputs are committed here as fixtures.

    python tests/golden/make_gmp_fixtures.py [family ...]

Families (inputs are regenerated from the seed by piplib_amd/synth.py; only outputs are stored):

  dense10, dense14  the dense large-coefficient batches of make_bigint_fixtures.py (all tableaux)
  wide128           ALL 1,000 tableaux of BASELINE configs[4]'s batch (128x256; the screened list of
                    tests/golden/bigint/wide128.json)
  param81..param84  the 24 random parametric problems each of
                    tests/test_gpu_golden.py::test_parametric_128bit_tree_vs_oracle128

Per problem: `status` (0 solution text, 1 void context, 2 the reference exit()ed), `pivots` (calls of
pivoter_gmp, interposed), `sha` = sha256 of the whitespace-squashed sol_edit text (`text` itself for
the small families and the first 24 of wide128), and the widths the reference's arithmetic reached
(oracle/ref_driver.c interposes libgmp's mul/add/sub): `entry_bits` over every tableau / cut /
context computation, `det_bits` for the single determinant of traiter.c:409-431.

What a 128-bit fixed-width run may be compared on:
  * `entry_bits` <= 127: no product, sum or difference of the run leaves the signed 128-bit range, so
    the 128-bit engine must reproduce text and pivot count bit for bit (`wrap128` false);
  * the GMP flavour keeps ONE unbounded determinant, the fixed-width flavours keep at most three limbs of
    Entier width and exit with "Integer overflow" when a fourth is needed (traiter.c:412-446).  Where the
    128-bit restatement stops that way and the reference's GMP build goes on, the record carries
    `limb_overflow128` true: the verdict differs by construction of the flavour, not by arithmetic.
"""
import hashlib
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import make_bigint_fixtures as mk  # noqa: E402
import pipbatch as pb  # noqa: E402
from piplib_amd import synth  # noqa: E402

PARAM = {"param81": (81, (5, 2, 9, 2), 60), "param82": (82, (6, 1, 10, 1), 200), "param83": (83, (4, 3, 8, 3), 30),
         "param84": (84, (8, 2, 12, 2), 25)}
FAMILIES = ["dense10", "dense14", "wide128"] + list(PARAM)
KEEP_TEXT = {"wide128": 24}   # full text only for the first N (the rest: hash); other families keep all
TIMEOUT = 40                  # seconds per problem; the reference may not terminate on random parametric inputs


def problems_of(fam):
    if fam in PARAM:
        seed, shape, cmax = PARAM[fam]
        return synth.random_problems(seed, 24, *shape, 1, cmax=cmax, bmax=4 * cmax), 0
    rows = mk.rows_full(fam)
    nvar = rows.shape[2] - 1
    return [synth.Problem(nvar, 0, rows.shape[1], 0, -1, 1, rows[b], np.zeros((0, 1), np.int64))
            for b in range(rows.shape[0])], pb.F_NOSIMPLIFY


def sha(text):
    return hashlib.sha256(pb.squash(text).encode()).hexdigest()


def run_one(args):
    exe, p, flags, timeout = args
    try:
        return pb.run_batch(exe, [p], flags, timeout=timeout).results[0]
    except subprocess.TimeoutExpired:
        return None


def make(fam):
    import concurrent.futures as cf
    probs, flags = problems_of(fam)
    t0 = time.time()
    ncpu = max(1, len(os.sched_getaffinity(0)))
    with cf.ThreadPoolExecutor(ncpu) as ex:
        gmp = list(ex.map(run_one, [(pb.REFPIP_GMP, p, flags, TIMEOUT) for p in probs]))
        o128 = list(ex.map(run_one, [(pb.ORACLEPIP128, p, flags, TIMEOUT) for p in probs]))
    keep = KEEP_TEXT.get(fam, len(probs))
    recs = []
    for i, (g, o) in enumerate(zip(gmp, o128)):
        if g is None:
            recs.append({"timeout": True})
            continue
        r = {"status": g.status, "pivots": g.pivots, "sha": sha(g.text), "entry_bits": g.entry_bits,
             "det_bits": g.det_bits, "wrap128": g.entry_bits > 127}
        if g.status == pb.ST_ABORT:
            r["abort_code"] = g.abort_code
        if o is not None and o.status == pb.ST_ABORT and g.status != pb.ST_ABORT and not r["wrap128"]:
            r["limb_overflow128"] = True
        if i < keep:
            r["text"] = g.text
        recs.append(r)
    done = [r for r in recs if "status" in r]
    print(f"{fam}: {len(recs)} problems in {time.time() - t0:.0f} s, {len(recs) - len(done)} timed out; "
          f"pivots {sum(r['pivots'] for r in done)}, entries beyond 2^63 in {sum(r['entry_bits'] > 63 for r in done)}, "
          f"beyond 2^127 in {sum(r['wrap128'] for r in done)}, limb verdict differs in "
          f"{sum(r.get('limb_overflow128', False) for r in done)}, statuses {sorted(set(r['status'] for r in done))}, "
          f"max det bits {max([r['det_bits'] for r in done] or [0])}")
    return {"family": fam, "made_by": "tests/golden/make_gmp_fixtures.py: oracle/_ref/refpip_gmp = the reference's five "
                                      "sources compiled with -DPIPLIB_INT_GMP (oracle/Makefile ref_gmp)",
            "timeout_seconds": TIMEOUT, "problems": recs}


def main():
    if not os.access(pb.REFPIP_GMP, os.X_OK):
        raise SystemExit("oracle/_ref/refpip_gmp missing: make -C oracle ref_gmp (needs /root/reference and gmp.h)")
    # the GMP driver on the reference's own .dat suite first: nothing overflows there, so it must print what the
    # int64 build of the same driver prints (which tests/test_oracle_golden.py pins to the .ll goldens)
    g = os.path.join(HERE, "test")
    n = 0
    for f in sorted(os.listdir(g)):
        if f.endswith(".dat"):
            got = subprocess.run([pb.REFPIP_GMP, "dat", os.path.join(g, f)], capture_output=True, timeout=120).stdout
            want = subprocess.run([pb.REFPIP, "dat", os.path.join(g, f)], capture_output=True, timeout=120).stdout
            assert got == want, f
            n += 1
    print(f"refpip_gmp == refpip (int64) on the reference's {n} .dat inputs")
    os.makedirs(os.path.join(HERE, "gmp"), exist_ok=True)
    for fam in ([a for a in sys.argv[1:] if a in FAMILIES] or FAMILIES):
        doc = make(fam)
        with open(os.path.join(HERE, "gmp", fam + ".json"), "w") as f:
            json.dump(doc, f, indent=0)
            f.write("\n")


if __name__ == "__main__":
    main()
