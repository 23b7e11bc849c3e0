"""-m gpu: the HIP path (through the C ABI in libpipamd.so) against the CPU oracle,
bit-exact: status, pivot count and every numerator/denominator of the solution."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


@pytest.mark.parametrize("seed,batch,nvar,ni,nq,kw", [
    (11, 64, 6, 8, 1, dict(nnz=3, cmax=3, x0max=5)),
    (12, 64, 6, 8, 0, dict(nnz=3, cmax=3, x0max=5)),
    (13, 64, 20, 16, 1, dict()),
    (14, 48, 63, 32, 0, dict()),      # BASELINE configs[1] shape: 32x64, rational
    (15, 48, 63, 32, 1, dict()),
    (16, 32, 127, 64, 1, dict()),     # BASELINE configs[2] shape: 64x128, integer + cuts
    (17, 16, 200, 90, 1, dict()),     # two 128-column chunks per row
])
def test_lexmin_batch_vs_oracle(seed, batch, nvar, ni, nq, kw):
    from gpu_common import compare
    from piplib_amd import synth
    rows = synth.lexmin_batch(seed, batch, nvar, ni, **kw)
    n, piv = compare(rows, nvar, 0, nq)
    assert n == batch and piv > 0


@pytest.mark.parametrize("seed,nvar,nparm,ni,nc,nq,deepest", [
    (21, 5, 2, 7, 2, 1, 0), (22, 5, 2, 7, 2, 0, 0), (23, 4, 3, 6, 3, 1, 0), (24, 3, 1, 5, 1, 1, 0),
    (25, 6, 0, 8, 0, 1, 1), (26, 5, 2, 7, 2, 1, 1), (27, 8, 2, 10, 1, 1, 0), (28, 6, 4, 8, 2, 1, 0),
])
def test_random_parametric_vs_oracle(seed, nvar, nparm, ni, nc, nq, deepest):
    """Host decision tree + HIP engine (pipamd_solve_tableau) vs the CPU oracle on random
    parametric problems: same text (splits, newparms, nil leaves), same pivot count, and the
    same abort verdict where the reference would exit."""
    import pipbatch as pb
    from piplib_amd import engine as eng
    from piplib_amd import synth
    import subprocess
    probs, results = [], []
    for p in synth.random_problems(seed, 40, nvar, nparm, ni, nc, nq):
        # some random parametric problems make the reference itself cut forever: keep the
        # ones the CPU oracle finishes promptly
        try:
            r = pb.run_batch(pb.ORACLEPIP, [p], pb.F_DEEPEST if deepest else 0, timeout=2).results[0]
        except subprocess.TimeoutExpired:
            continue
        if r.pivots <= 3000:
            probs.append(p)
            results.append(r)
    assert len(probs) >= 20

    class o:  # noqa: N801
        pass
    o.results = results
    e = eng.Engine(0)
    nontrivial = 0
    for i, (p, r) in enumerate(zip(probs, o.results)):
        try:
            text, piv = eng.solve_tableau(e, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, p.ineq, p.ctx,
                                          simplify=True, deepest_cut=bool(deepest))
        except eng.SolverError as ex:
            assert r.status == pb.ST_ABORT, (i, ex.status)
            continue
        assert r.status != pb.ST_ABORT, i
        want = "void" if r.status == pb.ST_VOID else pb.squash(r.text)
        assert pb.squash(text) == want, (i, text[:200], r.text[:200])
        assert piv == r.pivots, (i, piv, r.pivots)
        nontrivial += "if" in text or "newparm" in text
    assert nparm == 0 or nontrivial > 0


def _check_solutions(rows, nvar, num, den, st, integer):
    """Size-independent properties of a lexmin answer: every reported point satisfies every
    inequality of its tableau (exact rational arithmetic in int64/float-free form), is
    non-negative, and is integral when the integer solve was asked for."""
    import numpy as np
    from piplib_amd import engine as eng
    ok = st == eng.ST_SOLUTION
    x_num = num[:, :, 0].astype(object)           # nparm = 0: one value (the constant) per unknown
    x_den = den.astype(object)
    assert (den[ok] > 0).all()
    assert (num[ok][:, :, 0] >= 0).all()
    if integer:
        assert (num[ok][:, :, 0] % den[ok] == 0).all()
    # A x + c >= 0 with x_i = n_i / d_i: multiply by D = lcm(d) -- here: use exact Python ints
    bad = 0
    idx = np.nonzero(ok)[0]
    for b in idx[:: max(1, len(idx) // 400)]:      # exact check on up to ~400 tableaux
        A = rows[b, :, :nvar].astype(object)
        c = rows[b, :, nvar].astype(object)
        D = 1
        for d in x_den[b]:
            D = D * int(d) // __import__("math").gcd(D, int(d))
        xs = np.array([int(n) * (D // int(d)) for n, d in zip(x_num[b], x_den[b])], dtype=object)
        lhs = A.dot(xs) + c * D
        bad += int((lhs < 0).any())
    assert bad == 0
    return int(ok.sum())


@pytest.mark.parametrize("batch,nvar,ni,nq", [
    (10000, 127, 64, 1),   # BASELINE configs[2]: 10k x (64x128), integer solve with Gomory cuts
    (1000, 63, 32, 0),     # BASELINE configs[1]: 1k x (32x64), rational solve
])
def test_full_size_properties(batch, nvar, ni, nq):
    """BASELINE.json's full sizes: every tableau finishes, answers are feasible (and integral),
    a second solve of the same batch gives bit-identical results (idempotence / determinism
    across the round scheduler), and a random sample agrees exactly with the CPU oracle."""
    import numpy as np
    import torch
    from gpu_common import gpu_batch, oracle_batch, solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    rows = synth.lexmin_batch(1000, batch, nvar, ni)
    g = gpu_batch(rows, nvar, 0, nq)
    st = g.status.cpu().numpy()
    pv = g.pivots.cpu().numpy()
    num = g.sol_num.cpu().numpy()
    den = g.sol_den.cpu().numpy()
    assert np.isin(st, [eng.ST_SOLUTION, eng.ST_NIL]).all(), np.unique(st, return_counts=True)
    assert _check_solutions(rows, nvar, num, den, st, bool(nq)) > 0.9 * batch
    # idempotence: reload + resolve with a different round length
    g.e.set_round_pivots(13)
    g.load()
    g.solve()
    g.fetch()
    torch.cuda.synchronize()
    assert (g.status.cpu().numpy() == st).all() and (g.pivots.cpu().numpy() == pv).all()
    assert (g.sol_num.cpu().numpy() == num).all() and (g.sol_den.cpu().numpy() == den).all()
    # exact agreement with the oracle on a sample
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(batch, size=min(200, batch), replace=False))
    o = oracle_batch(rows[pick], nvar, 0, nq)
    for b, r in zip(pick, o.results):
        assert pv[b] == r.pivots, b
        want = pb.squash(r.text)
        got = "()" if st[b] == eng.ST_NIL else pb.squash(solution_text(num[b], den[b]))
        assert got == want, b
