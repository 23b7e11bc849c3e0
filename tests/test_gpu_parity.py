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
    (18, 4, 60, 1900, 0, dict()),     # many rows: close to the 2048-slot limit, LDS image ~90 KB
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


def _gpu128(rows, nvar, nq, cap_cuts):
    import torch
    from piplib_amd import engine as eng
    e = eng.Engine(0)
    b = eng.Batch(e, rows, nvar, 0, tflags=eng.T_INT if nq else 0, cap_cuts=cap_cuts, entier_bits=128)
    b.load()
    b.solve()
    b.fetch()
    torch.cuda.synchronize()
    return b


def test_int128_equals_int64_on_small_entries():
    """128-bit Entier variant == the int64 oracle when nothing overflows."""
    import numpy as np
    from gpu_common import oracle_batch, solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    rows = synth.lexmin_batch(31, 48, 40, 24)
    o = oracle_batch(rows, 40, 0, 1)
    g = _gpu128(rows, 40, 1, 96)
    st, pv = g.status.cpu().numpy(), g.pivots.cpu().numpy()
    num, den = eng.wide_to_int(g.sol_num.cpu().numpy()), eng.wide_to_int(g.sol_den.cpu().numpy())
    for b, r in enumerate(o.results):
        assert pv[b] == r.pivots, b
        got = "()" if st[b] == eng.ST_NIL else pb.squash(solution_text(num[b], den[b]))
        assert st[b] in (eng.ST_SOLUTION, eng.ST_NIL) and got == pb.squash(r.text), b


@pytest.mark.parametrize("seed,nvar,ni,cmax", [(3, 10, 10, 40), (4, 14, 12, 60)])
def test_int128_overflow_safe_path(seed, nvar, ni, cmax):
    """Dense large-coefficient tableaux: the int64 reference aborts ("Integer overflow") on many
    of them; the 128-bit engine must agree with the 128-bit oracle (same algorithm on __int128)
    in status, pivot count and every numerator/denominator."""
    import numpy as np
    from gpu_common import solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    rows = synth.dense_batch(seed, 48, nvar, ni, cmax)
    probs = [synth.Problem(nvar, 0, ni, 0, -1, 1, rows[b], np.zeros((0, 1), np.int64)) for b in range(rows.shape[0])]
    o64 = pb.run_batch(pb.ORACLEPIP, probs, pb.F_NOSIMPLIFY)
    o128 = pb.run_batch(pb.ORACLEPIP128, probs, pb.F_NOSIMPLIFY)
    assert sum(r.status == pb.ST_ABORT for r in o64.results) > 5      # int64 really overflows here
    g = _gpu128(rows, nvar, 1, 700)
    st, pv = g.status.cpu().numpy(), g.pivots.cpu().numpy()
    num, den = eng.wide_to_int(g.sol_num.cpu().numpy()), eng.wide_to_int(g.sol_den.cpu().numpy())
    checked = 0
    for b, r in enumerate(o128.results):
        if st[b] == eng.ST_CAPACITY:
            continue
        if r.status == pb.ST_ABORT:
            assert st[b] == eng.ST_OVERFLOW, b
            continue
        assert pv[b] == r.pivots, (b, pv[b], r.pivots)
        got = "()" if st[b] == eng.ST_NIL else pb.squash(solution_text(num[b], den[b]))
        assert got == pb.squash(r.text), b
        checked += 1
    assert checked >= 30


def test_full_size_int128_config():
    """BASELINE configs[4]: 1k-batch 128x256 tableaux on the 128-bit Entier path.  Entries of the
    sparse generator stay small, so the int64 oracle is the checker for a sample; all tableaux
    must finish and be feasible + integral."""
    import numpy as np
    from gpu_common import oracle_batch, solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    nvar, ni, batch = 255, 128, 1000
    rows = synth.lexmin_batch(77, batch, nvar, ni)
    g = _gpu128(rows, nvar, 1, None)
    st, pv = g.status.cpu().numpy(), g.pivots.cpu().numpy()
    assert np.isin(st, [eng.ST_SOLUTION, eng.ST_NIL]).all(), np.unique(st, return_counts=True)
    num = eng.wide_to_int(g.sol_num.cpu().numpy())
    den = eng.wide_to_int(g.sol_den.cpu().numpy())
    ok = st == eng.ST_SOLUTION
    assert all((d > 0).all() and (n[:, 0] >= 0).all() and all(int(a) % int(b) == 0 for a, b in zip(n[:, 0], d))
               for n, d in zip(num[ok][:100], den[ok][:100]))
    pick = np.arange(0, batch, 25)
    o = oracle_batch(rows[pick], nvar, 0, 1)
    for b, r in zip(pick, o.results):
        assert pv[b] == r.pivots, b
        got = "()" if st[b] == eng.ST_NIL else pb.squash(solution_text(num[b], den[b]))
        assert got == pb.squash(r.text), b


def test_many_parametric_problems_multithreaded():
    """pipamd_solve_tableaux (8 host threads / streams) and pipamd_solve_tableaux_lockstep (one
    clone/patch/launch/gather sequence per step for the whole batch) give exactly the answers of
    the one-at-a-time entry point (which the other tests pin to the oracle)."""
    from piplib_amd import engine as eng, synth
    probs = [p for seed, shape in ((41, (5, 2, 7, 2)), (42, (4, 3, 6, 3)), (43, (6, 1, 8, 1)))
             for p in synth.random_problems(seed, 30, *shape, 1)]
    import pipbatch as pb
    import subprocess
    keep = []
    for p in probs:  # drop the few the reference itself never finishes
        try:
            if pb.run_batch(pb.ORACLEPIP, [p], timeout=2).results[0].pivots <= 3000:
                keep.append(p)
        except subprocess.TimeoutExpired:
            pass
    e = eng.Engine(0)
    many = eng.solve_tableaux(e, keep, nthreads=8)
    # the lock-step scheduler (one launch per step for all problems) must agree entry by entry
    assert eng.solve_tableaux(e, keep, lockstep=True) == many
    for p, (text, rc, st, piv) in zip(keep, many):
        try:
            t1, p1 = eng.solve_tableau(e, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, p.ineq, p.ctx)
            assert rc == 0 and text == t1 and piv == p1
        except eng.SolverError as ex:
            assert rc == -5 and st == ex.status


@pytest.mark.parametrize("seed", [51, 52, 53])
def test_engine_variants_agree(seed):
    """One batch through every engine configuration -- 1 or 4 waves per tableau, short rounds,
    row skipping off, 128-bit entries -- must give bit-identical statuses, pivot counts and
    solutions (and match the oracle, checked on the first configuration)."""
    import numpy as np
    import torch
    from gpu_common import oracle_batch, solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    rng = np.random.default_rng(seed)
    nvar, ni = int(rng.integers(5, 90)), int(rng.integers(4, 48))
    rows = synth.lexmin_batch(seed, 40, nvar, ni, nnz=int(rng.integers(2, 6)), cmax=int(rng.integers(2, 9)))
    nq = int(seed % 2 == 1)
    outs = []
    for waves, rnd, extra, bits in [(4, 0, 0, 64), (1, 0, 0, 64), (1, 5, 0, 64), (4, 3, eng.T_NOSKIP, 64), (4, 0, 0, 128)]:
        e = eng.Engine(0)
        e.set_waves_per_job(waves)
        if rnd:
            e.set_round_pivots(rnd)
        b = eng.Batch(e, rows, nvar, 0, tflags=(eng.T_INT if nq else 0) | extra, cap_cuts=200, entier_bits=bits)
        b.load()
        b.solve()
        b.fetch()
        torch.cuda.synchronize()
        num, den = b.sol_num.cpu().numpy(), b.sol_den.cpu().numpy()
        if bits == 128:
            num, den = eng.wide_to_int(num), eng.wide_to_int(den)
        outs.append((b.status.cpu().numpy(), b.pivots.cpu().numpy(), num.astype(object), den.astype(object)))
    for o in outs[1:]:
        assert (o[0] == outs[0][0]).all() and (o[1] == outs[0][1]).all()
        assert (o[2] == outs[0][2]).all() and (o[3] == outs[0][3]).all()
    ora = oracle_batch(rows, nvar, 0, nq)
    st, pv, num, den = outs[0]
    for b_, r in enumerate(ora.results):
        if st[b_] == eng.ST_CAPACITY:
            continue
        assert pv[b_] == r.pivots
        got = "()" if st[b_] == eng.ST_NIL else pb.squash(solution_text(num[b_], den[b_]))
        assert got == pb.squash(r.text)


def test_pip_solve_options_fuzz():
    """pipamd_pip_solve with Maximize / Urs_unknowns / Urs_parms / rational on random small
    PolyLib matrices vs the oracle's pip front end (same text, incl. 'void')."""
    import subprocess
    import tempfile
    import numpy as np
    import pipbatch as pb
    from datfile import matrix_text
    from piplib_amd import engine as eng
    rng = np.random.default_rng(9)
    e = eng.Engine(0)
    checked = 0
    for trial in range(60):
        nn, npar = int(rng.integers(1, 4)), int(rng.integers(0, 3))
        nrow, ncrow = int(rng.integers(1, 6)), int(rng.integers(0, 3))
        dom = rng.integers(-3, 4, size=(nrow, nn + npar + 2)).astype(np.int64)
        dom[:, 0] = rng.random(nrow) < 0.85          # mostly inequalities, some equalities
        dom[:, -1] = rng.integers(-6, 9, size=nrow)
        ctx = rng.integers(-2, 3, size=(ncrow, npar + 2)).astype(np.int64)
        if ncrow:
            ctx[:, 0] = 1
            ctx[:, -1] = rng.integers(0, 9, size=ncrow)
        opts = {}
        k = trial % 5
        if k == 1:
            opts["Maximize"] = 1
        elif k == 2:
            opts["Urs_unknowns"] = 1
        elif k == 3 and npar:
            opts["Urs_parms"] = 1
        elif k == 4:
            opts["Nq"] = 0
        words = {"Maximize": "Maximize", "Urs_unknowns": "Urs_unknowns", "Urs_parms": "Urs_parms"}
        txt = matrix_text(ctx) + "\n-1\n\n" + matrix_text(dom) + "\n" + \
            "".join(words[o] + "\n" for o in opts if o in words) + ("Rational\n" if opts.get("Nq") == 0 else "")
        try:
            p = subprocess.run([pb.ORACLEPIP, "pip"], input=txt.encode(), capture_output=True, timeout=3)
        except subprocess.TimeoutExpired:
            continue
        if p.returncode != 0:
            continue
        want = pb.squash(p.stdout.decode())
        try:
            text, _ = eng.pip_solve(e, dom, ctx, -1, **opts)
        except eng.SolverError:
            continue
        head = ("[PIP2-like future input] Please enter:\n- the context matrix,\n" + matrix_text(ctx) +
                "- the bignum column (start at 0, -1 if no bignum),\n-1\n- the constraint matrix.\n" +
                matrix_text(dom) + "\n")
        assert pb.squash(head + text) == want, (trial, opts, text[:200])
        checked += 1
    assert checked >= 40


def test_largest_supported_shape():
    """Engine limits (the reference's MAXCOL: fewer than 512 columns; 1024 logical rows): a
    510-unknown tableau with hundreds of rows must run (large dynamic LDS image) and agree with
    the oracle; with 512 columns both stop with "Too many variables" (integrer.c:324)."""
    from gpu_common import compare
    from piplib_amd import synth
    rows = synth.lexmin_batch(61, 6, 510, 300, nnz=3, cmax=3, x0max=4)
    n, piv = compare(rows, 510, 0, 1, cap_cuts=200)
    assert n == 6 and piv > 0
    rows = synth.lexmin_batch(62, 4, 511, 40, nnz=3, cmax=3, x0max=4)
    compare(rows, 511, 0, 1, cap_cuts=60)


@pytest.mark.parametrize("seed,nvar,ni", [(71, 10, 12), (72, 30, 20), (73, 127, 64)])
def test_deepest_cut_on_device(seed, nvar, ni):
    """PIPAMD_T_DEEPEST (the reference's -d / Deepest_cut option, integrer.c:417-438) entirely in
    the kernel, batch API, vs the oracle run with the same option."""
    import numpy as np
    import torch
    from gpu_common import solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    rows = synth.lexmin_batch(seed, 32, nvar, ni)
    probs = [synth.Problem(nvar, 0, ni, 0, -1, 1, rows[b], np.zeros((0, 1), np.int64)) for b in range(32)]
    o = pb.run_batch(pb.ORACLEPIP, probs, pb.F_NOSIMPLIFY | pb.F_DEEPEST)
    plain = pb.run_batch(pb.ORACLEPIP, probs, pb.F_NOSIMPLIFY)
    e = eng.Engine(0)
    b = eng.Batch(e, rows, nvar, 0, tflags=eng.T_INT | 512, cap_cuts=250)
    b.load()
    b.solve()
    b.fetch()
    torch.cuda.synchronize()
    st, pv = b.status.cpu().numpy(), b.pivots.cpu().numpy()
    num, den = b.sol_num.cpu().numpy(), b.sol_den.cpu().numpy()
    for k, r in enumerate(o.results):
        if st[k] == eng.ST_CAPACITY:
            continue
        assert r.status == pb.ST_OK and st[k] in (eng.ST_SOLUTION, eng.ST_NIL)
        assert pv[k] == r.pivots, (k, pv[k], r.pivots)
        got = "()" if st[k] == eng.ST_NIL else pb.squash(solution_text(num[k], den[k]))
        assert got == pb.squash(r.text), k
    # the option really changes the pivot sequence on this workload
    assert sum(a.pivots != c.pivots for a, c in zip(o.results, plain.results)) > 0


@pytest.mark.parametrize("nvar,ni", [(1, 1), (2, 1), (1, 5), (3, 2), (64, 1), (129, 3)])
def test_degenerate_shapes(nvar, ni):
    """Tiny and lopsided tableaux (incl. all-zero rows, infeasible and unbounded-looking ones)."""
    import numpy as np
    from gpu_common import compare
    rng = np.random.default_rng(nvar * 100 + ni)
    rows = rng.integers(-3, 4, size=(24, ni, nvar + 1)).astype(np.int64)
    rows[0] = 0                      # all-zero tableau
    rows[1, :, :nvar] = 0            # constants only
    rows[1, :, nvar] = -1            # ... and infeasible
    for nq in (0, 1):
        n, _ = compare(rows, nvar, 0, nq, cap_cuts=64)
        assert n == 24
