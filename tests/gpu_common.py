"""Helpers for the -m gpu parity tests: HIP engine (through the C ABI) vs the CPU oracle."""
import math

import numpy as np

import pipbatch as pb
from piplib_amd import engine as eng
from piplib_amd import synth


def frac_text(N, D):
    """sol.c:343-374: value N/D printed after dividing both by gcd(N, D)."""
    N, D = int(N), int(D)
    d = math.gcd(N, D)
    if d == D:
        return f" {N // d}"
    return f" {N // d}/{D // d}"


def solution_text(num, den):
    """sol_edit text of a List of nvar Forms (traiter.c:255-271 + sol.c:335-378)."""
    out = ["(list "]
    for i in range(num.shape[0]):
        out.append("#[" + "".join(frac_text(n, den[i]) for n in num[i]) + "]\n")
    out.append(")\n")
    return "".join(out)


def oracle_batch(rows, nvar, nparm, nq, bigparm=-1):
    probs = [synth.Problem(nvar, nparm, rows.shape[1], 0, bigparm, nq, rows[b], np.zeros((0, nparm + 1), np.int64))
             for b in range(rows.shape[0])]
    return pb.run_batch(pb.ORACLEPIP, probs, pb.F_NOSIMPLIFY)


def gpu_batch(rows, nvar, nparm, nq, bigparm=-1, cap_cuts=None, iter_limit=None):
    import torch
    e = eng.Engine(0)
    if iter_limit:
        e.set_iter_limit(iter_limit)
    b = eng.Batch(e, rows, nvar, nparm, bigparm=bigparm, tflags=eng.T_INT if nq else 0, cap_cuts=cap_cuts)
    b.load()
    b.solve()
    b.fetch()
    torch.cuda.synchronize()
    return b


def compare(rows, nvar, nparm, nq, bigparm=-1, cap_cuts=None):
    """Returns (n_checked, total_pivots); asserts bit-exact agreement with the oracle."""
    o = oracle_batch(rows, nvar, nparm, nq, bigparm)
    g = gpu_batch(rows, nvar, nparm, nq, bigparm, cap_cuts)
    st = g.status.cpu().numpy()
    pv = g.pivots.cpu().numpy()
    num = g.sol_num.cpu().numpy()
    den = g.sol_den.cpu().numpy()
    for b, r in enumerate(o.results):
        if r.status == pb.ST_ABORT:
            # oracle abort codes (pip_oracle.h): 2 = "Integer overflow", 4 = "Too many variables"
            want_st = {2: eng.ST_OVERFLOW, 4: eng.ST_MAXCOL}.get(r.abort_code, eng.ST_OVERFLOW)
            assert st[b] == want_st, (b, st[b], r.abort_code)
            continue
        want = pb.squash(r.text)
        if want == "()":
            assert st[b] == eng.ST_NIL, (b, st[b])
        else:
            assert st[b] == eng.ST_SOLUTION, (b, st[b], want[:80])
            got = pb.squash(solution_text(num[b], den[b]))
            assert got == want, (b, got[:200], want[:200])
        assert pv[b] == r.pivots, (b, pv[b], r.pivots)
    return len(o.results), int(pv.sum())
