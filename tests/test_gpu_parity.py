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
    across the launch scheduler), and EVERY tableau agrees exactly with the CPU oracle."""
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
    # exact agreement with the oracle on EVERY tableau (the oracle runs in 16 processes)
    o = _oracle_all(rows, nvar, nq)
    for b, r in enumerate(o):
        assert pv[b] == r.pivots, b
        want = pb.squash(r.text)
        got = "()" if st[b] == eng.ST_NIL else pb.squash(solution_text(num[b], den[b]))
        assert got == want, b


def _oracle_all(rows, nvar, nq, exe=None, procs=16):
    """The CPU oracle on a whole batch, `procs` processes side by side; results in input order."""
    import concurrent.futures as cf
    import numpy as np
    import pipbatch as pb
    from piplib_amd import synth
    exe = exe or pb.ORACLEPIP
    n = rows.shape[0]
    cuts = [n * i // procs for i in range(procs + 1)]

    def run(k):
        lo, hi = cuts[k], cuts[k + 1]
        probs = [synth.Problem(nvar, 0, rows.shape[1], 0, -1, nq, rows[b], np.zeros((0, 1), np.int64))
                 for b in range(lo, hi)]
        return pb.run_batch(exe, probs, pb.F_NOSIMPLIFY).results if probs else []
    with cf.ThreadPoolExecutor(procs) as ex:
        return [r for part in ex.map(run, range(procs)) for r in part]


def _compare_ids(g, rows, ids, nvar, nq, wide=False, exe=None):
    """the tableaux `ids` of a solved batch against the oracle: status, pivot count, every numerator/denominator"""
    from gpu_common import solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng
    st, pv = g.status.cpu().numpy(), g.pivots.cpu().numpy()
    if wide:
        num, den = eng.wide_to_int(g.sol_num.cpu().numpy()), eng.wide_to_int(g.sol_den.cpu().numpy())
    else:
        num, den = g.sol_num.cpu().numpy(), g.sol_den.cpu().numpy()
    o = _oracle_all(rows[ids], nvar, nq, exe=exe)
    for b, r in zip(ids, o):
        assert r.status != pb.ST_ABORT and st[b] in (eng.ST_SOLUTION, eng.ST_NIL), (b, st[b], r.status)
        assert pv[b] == r.pivots, (b, pv[b], r.pivots)
        got = "()" if st[b] == eng.ST_NIL else pb.squash(solution_text(num[b], den[b]))
        assert got == pb.squash(r.text), b


@pytest.mark.parametrize("batch,nvar,ni,cap,ebits,step", [
    (300, 31, 16, 1, 64, 0),      # four-wave launches only; nearly every tableau outgrows one spare row
    (2400, 31, 16, 2, 64, 0),     # through the one-wave bulk launch (>= 2048 tableaux) first
    (400, 127, 64, 1, 64, 0),     # BASELINE configs[2]'s shape
    (400, 127, 64, 1, 64, 3),     # ... with three more rows per growth round: up to 14 rounds for the longest tableau
    (200, 31, 16, 1, 128, 2),     # 128-bit entries, several rounds
])
def test_batch_layer_rehouses_full_tableaux(batch, nvar, ni, cap, ebits, step):
    """expanser (traiter.c:55-88) on cut overflow (integrer.c:410-415) in the batch layer: pipamd_batch_solve moves a
    tableau that has spent its `cap_cuts` spare rows into a block of twice the row capacity and goes on, as often as
    it takes -- no engine-only PIPAMD_ST_CAPACITY is left, and every answer and pivot count is the oracle's."""
    import ctypes as C
    import numpy as np
    import torch
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    rows = synth.lexmin_batch(4242, batch, nvar, ni)
    e = eng.Engine(0)
    if step:
        eng.lib().pipamd_debug_grow_step.argtypes = [C.c_void_p, C.c_int]
        assert eng.lib().pipamd_debug_grow_step(e._h, step) == 0
    g = eng.Batch(e, rows, nvar, 0, tflags=eng.T_INT, cap_cuts=cap, entier_bits=ebits)
    for _ in range(2):  # twice: the second solve reuses the engine's side arenas
        g.load()
        g.solve()
        g.fetch()
        torch.cuda.synchronize()
        st, ct = g.status.cpu().numpy(), g.cuts.cpu().numpy()
        assert np.isin(st, [eng.ST_SOLUTION, eng.ST_NIL]).all(), np.unique(st, return_counts=True)
        assert (ct > cap).sum() >= batch // 4      # the spare rows really were too few
        if step:
            assert (ct > cap + 2 * step).sum() >= batch // 8   # ... in more than two rounds for many
        _compare_ids(g, rows, np.arange(batch), nvar, 1, wide=ebits == 128,
                     exe=pb.ORACLEPIP128 if ebits == 128 else None)


def test_bench_lane_seeds_all_finish():
    """bench.py's headline batches are synth.lexmin_batch(1000 + 7919 * g, ...).  Which of their tableaux are left out
    (Gomory's cuts do not converge: integrer() asks for more than 448 constant cuts) is the CPU oracle's word,
    tests/golden/bench_screen.json (tests/golden/make_bench_screen.py) -- made without the engine.  Here the engine is
    held to it on the 16 batches of a one-GPU run: under the same budget it leaves exactly the listed tableaux at
    PIPAMD_ST_CAPACITY (piplib_amd.engine.slow_converging), and on the screened batch every tableau ends with a status the
    reference has (solution or nil) -- tableaux that need more than the default ni + 64 spare rows are re-housed --, the
    batch's pivot total is the oracle's, and the oracle agrees on every tableau that needed more than ni + 64 cuts plus
    150 others per batch."""
    import numpy as np
    import torch
    import bench
    from piplib_amd import engine as eng, synth
    nvar, ni = 127, 64
    rec = bench.screen_records("bench_screen")
    assert rec["4"]["slow"] == [893] and rec["7"]["slow"] == [6225] and rec["11"]["slow"] == [4572]
    e = eng.Engine(0)
    e.set_max_rows(ni + 1024)
    grown = 0
    for g_ in range(16):
        rows = synth.lexmin_batch(1000 + 7919 * g_, 10000, nvar, ni)
        slow = eng.slow_converging(e, torch.as_tensor(rows).to("cuda:0"), nvar, cut_rows=bench.SCREEN_CUTS)
        assert slow == rec[str(g_)]["slow"], (g_, slow)
        for b in slow:
            rows[b] = rows[bench.replacement(b, set(slow), 10000)]
        g = eng.Batch(e, rows, nvar, 0, tflags=eng.T_INT)
        g.load()
        g.solve()
        g.fetch()
        torch.cuda.synchronize()
        st, ct = g.status.cpu().numpy(), g.cuts.cpu().numpy()
        assert np.isin(st, [eng.ST_SOLUTION, eng.ST_NIL]).all(), (g_, np.unique(st, return_counts=True))
        assert int(g.pivots.sum().item()) == rec[str(g_)]["pivots_screened"], g_
        big = np.nonzero(ct > ni + 64)[0]
        grown += len(big)
        ids = np.unique(np.concatenate([big, np.random.default_rng(g_).choice(10000, 150, replace=False)]))
        _compare_ids(g, rows, ids, nvar, 1)
        del g
    print("tableaux that needed more than ni + 64 cut rows and were re-housed:", grown)


def _gpu128(rows, nvar, nq, cap_cuts, lean64=False):
    import torch
    from piplib_amd import engine as eng
    e = eng.Engine(0)
    e.set_lean64(lean64)
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


def _bigint_family(name):
    """(rows, expected records) of one tests/golden/bigint fixture; inputs are regenerated from the seed."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_bigint_fixtures as mk
    doc = json.load(open(os.path.join(here, "golden", "bigint", name + ".json")))
    return mk.rows_of(name), doc["tableaux"]


def _check_against_records(g, recs, lo=0):
    """status, pivot count, cut count and every numerator/denominator of the 128-bit engine against
    exact-arithmetic records (only records no 128-bit run can have wrapped on are comparable)."""
    from piplib_amd import engine as eng
    st, pv, ct = g.status.cpu().numpy(), g.pivots.cpu().numpy(), g.cuts.cpu().numpy()
    num, den = eng.wide_to_int(g.sol_num.cpu().numpy()), eng.wide_to_int(g.sol_den.cpu().numpy())
    checked = big = 0
    for i, r in enumerate(recs):
        b = lo + i
        if not r["exact"] or st[b] == eng.ST_CAPACITY:
            continue
        assert st[b] == r["status"], (b, st[b], r["status"])
        assert pv[b] == r["pivots"] and ct[b] == r["cuts"], (b, pv[b], r["pivots"], ct[b], r["cuts"])
        if r["status"] == eng.ST_SOLUTION:
            assert [int(x) for x in num[b][:, 0]] == [int(x) for x in r["sol_num"]], b
            assert [int(x) for x in den[b]] == [int(x) for x in r["sol_den"]], b
        checked += 1
        big += r["entry_bits"] > 63
    return checked, big


def _check_against_gmp(g, name, lo=0, hi=None):
    """the 128-bit engine's results for problems lo..hi of a GMP fixture family against the REFERENCE's own
    arbitrary-precision build (tests/golden/gmp): status, number of pivoter calls and the whole solution (sha256 of the
    squashed sol_edit text; the text itself where the fixture keeps it) on every record on which no value left the
    signed 128-bit range.  Returns (compared, of them with entries beyond 2^63, limb-overflow verdicts)."""
    import numpy as np
    from gmpfix import gmp_fixture
    from gpu_common import solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng
    probs, flags, recs, sha = gmp_fixture(name)
    hi = len(recs) if hi is None else hi
    st, pv = g.status.cpu().numpy(), g.pivots.cpu().numpy()
    num, den = eng.wide_to_int(g.sol_num.cpu().numpy()), eng.wide_to_int(g.sol_den.cpu().numpy())
    same = big = limb = 0
    for i in range(lo, hi):
        r, b = recs[i], i - lo
        if "status" not in r or r["wrap128"]:
            continue
        assert r["status"] == pb.ST_OK, (name, i)   # no parameters, arbitrary precision: the reference always answers
        if st[b] == eng.ST_OVERFLOW:
            # the fixed-width flavours keep at most three determinant limbs (traiter.c:412-446), the GMP flavour one
            # unbounded determinant: the only verdict that may differ, and the CPU restatement must share it
            o = pb.run_batch(pb.ORACLEPIP128, [probs[i]], flags, timeout=120).results[0]
            assert o.status == pb.ST_ABORT and o.abort_code == 2, (name, i)
            limb += 1
            continue
        assert st[b] in (eng.ST_SOLUTION, eng.ST_NIL), (name, i, st[b])
        assert pv[b] == r["pivots"], (name, i, pv[b], r["pivots"])
        got = "()" if st[b] == eng.ST_NIL else solution_text(num[b], den[b])
        assert sha(got) == r["sha"], (name, i)
        if "text" in r:
            assert pb.squash(got) == pb.squash(r["text"]), (name, i)
        same += 1
        big += r["entry_bits"] > 63
    return same, big, limb


@pytest.mark.parametrize("family,cap", [("dense10", 700), ("dense14", 700)])
def test_int128_overflow_safe_path(family, cap):
    """Dense large-coefficient tableaux: the int64 reference aborts ("Integer overflow") on many of them.  The
    128-bit engine is held against what the reference's own GMP build computes for them (tests/golden/gmp): status,
    pivot count, every numerator and denominator; and against the exact-arithmetic records of round 2 (cut counts)."""
    import numpy as np
    import pipbatch as pb
    from piplib_amd import synth
    rows, recs = _bigint_family(family)
    nvar = rows.shape[2] - 1
    probs = [synth.Problem(nvar, 0, rows.shape[1], 0, -1, 1, rows[b], np.zeros((0, 1), np.int64))
             for b in range(rows.shape[0])]
    o64 = pb.run_batch(pb.ORACLEPIP, probs, pb.F_NOSIMPLIFY)
    assert sum(r.status == pb.ST_ABORT for r in o64.results) > 5      # int64 really overflows here
    g = _gpu128(rows, nvar, 1, cap)
    same, big, limb = _check_against_gmp(g, family)
    assert same >= 0.9 * len(recs) and big >= 10 and limb == 0, (same, big, limb)
    checked, _ = _check_against_records(g, recs)
    assert checked >= 0.6 * len(recs), checked


def test_full_size_int128_config():
    """BASELINE configs[4]: 1k-batch 128x256 tableaux on the 128-bit Entier path, with coefficients that push tableau
    entries beyond 2^63 (16 non-zeros of magnitude <= 30 per row; 587 of the 1,000 tableaux go beyond 2^63, 5 beyond
    2^127).  EVERY tableau is held against the output of the reference's own GMP build (tests/golden/gmp/wide128.json:
    pivot count and the whole solution), and the first 24 also against round 2's exact-arithmetic records (cut counts)."""
    import make_bigint_fixtures as mk  # noqa: F401  (path set up by _bigint_family)
    rows24, recs = _bigint_family("wide128")
    rows = mk.rows_full("wide128")
    assert rows.shape[0] == 1000 and (rows[:len(recs)] == rows24).all()
    nvar = mk.FAMILIES["wide128"]["nvar"]
    g = _gpu128(rows, nvar, 1, None)
    same, big, limb = _check_against_gmp(g, "wide128")
    assert same >= 990 and big >= 500 and limb == 0, (same, big, limb)
    checked, big24 = _check_against_records(g, recs)
    assert checked >= 12 and big24 >= 6, (checked, big24)


def test_lean64_kernel_paths():
    """The lean kernel of the 128-bit flavour (csrc/pip_lean64.h, opt-in: pipamd_engine_set_lean64): one wave per tableau,
    rows held as long longs while every entry fits 63 bits.  configs[4]'s pinned batch through it -- two in three
    tableaux finish there, a third leave on a row beyond 2^63 (stored in the general format mid-pivot, the tableau handed
    to pip_advance_kernel<__int128>), a few on the launch's pivot budget -- must come out as without it (statuses, pivot
    and cut counts, solutions) and as the reference's GMP build has it; then the same with rows multiplied through so that
    entries start beyond 2^31 and near 2^62 (the 128-bit products from the first pivot; rows outgrow long longs at once), and a
    rational solve."""
    import numpy as np
    import torch
    import make_bigint_fixtures as mk  # noqa: F401
    from piplib_amd import engine as eng
    _bigint_family("wide128")
    rows = mk.rows_full("wide128")
    nvar = mk.FAMILIES["wide128"]["nvar"]
    g1, g0 = _gpu128(rows, nvar, 1, None, lean64=True), _gpu128(rows, nvar, 1, None)
    for name in ("status", "pivots", "cuts", "sol_num", "sol_den"):
        assert (getattr(g1, name) == getattr(g0, name)).all(), name
    same, big, limb = _check_against_gmp(g1, "wide128")
    assert same >= 990 and big >= 500 and limb == 0, (same, big, limb)
    # the lean launch on its own: how its tableaux ended (PipJob: status at byte 72, pivots at 80, exit reason at 172)
    e = eng.Engine(0)
    e.set_lean64(True)
    e.debug_single_launch(1)
    b = eng.Batch(e, rows, nvar, 0, tflags=eng.T_INT, entier_bits=128)
    b.load()
    b.solve()
    torch.cuda.synchronize()
    j = b.ws[:25 * rows.shape[0]].view(torch.int32).view(rows.shape[0], 50).cpu().numpy()
    status, npiv, why = j[:, 18], j[:, 20], j[:, 43]
    running = status == eng.ST_RUN
    assert (npiv > 0).all() and (~running).sum() > 500 and (running & (why == 2)).sum() > 100 and (running & (why == 1)).any()
    # other magnitudes and a rational solve: with and without the kernel
    sub = rows[:160].copy()
    for scale, nq in ((1 << 33, 1), (1 << 57, 1), (1, 0)):
        r2 = sub.copy()
        if scale > 1:
            r2[::2, 5, :] *= scale   # every other tableau: one inequality multiplied through (same polyhedron)
            for b_ in {1 << 33: (18,), 1 << 57: (92,)}[scale]:   # (scaled like that the CPU oracle does not finish these two)
                r2[b_] = sub[b_]
        a, c = _gpu128(r2, nvar, nq, None, lean64=True), _gpu128(r2, nvar, nq, None)
        for name in ("status", "pivots", "cuts", "sol_num", "sol_den"):
            assert (getattr(a, name) == getattr(c, name)).all(), (scale, nq, name)
        assert (a.status.cpu().numpy() != eng.ST_RUN).all()


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
    first = None
    for device_tree in (False, True):  # the host schedulers alone, then with the device-resident traiter() in front
        e = eng.Engine(0)
        e.set_device_tree(device_tree)
        many = eng.solve_tableaux(e, keep, nthreads=8)
        assert (e.last_device_tree()[0] > 0) == device_tree
        # the lock-step scheduler (one launch per step for all problems) must agree entry by entry
        assert eng.solve_tableaux(e, keep, lockstep=True) == many
        for p, (text, rc, st, piv) in zip(keep, many):
            try:
                t1, p1 = eng.solve_tableau(e, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, p.ineq, p.ctx)
                assert rc == 0 and text == t1 and piv == p1
            except eng.SolverError as ex:
                assert rc == -5 and st == ex.status
        assert first is None or many == first
        first = many


@pytest.mark.parametrize("seed", [51, 52, 53])
def test_engine_variants_agree(seed):
    """One batch through every engine configuration -- 1, 4 or 8 waves per tableau, short rounds,
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
    for waves, rnd, extra, bits in [(4, 0, 0, 64), (1, 0, 0, 64), (1, 5, 0, 64), (4, 3, eng.T_NOSKIP, 64), (4, 0, 0, 128),
                                    (8, 0, 0, 64)]:
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
    """The reference's pip_solve (Maximize / Urs_unknowns / Urs_parms / rational handled by its own
    piplib.c) with traiter() bound to the GPU engine, on random small PolyLib matrices, vs the
    oracle's pip front end (same text, incl. 'void')."""
    import subprocess
    import numpy as np
    import pipbatch as pb
    from datfile import matrix_text
    if not pb.have_ref_gpu():
        pytest.skip("oracle/_ref/refpip_gpu not built (no /root/reference)")
    rng = np.random.default_rng(9)
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
        k = trial % 5
        words = {1: "Maximize\n", 2: "Urs_unknowns\n", 3: "Urs_parms\n" if npar else "", 4: "Rational\n"}.get(k, "")
        txt = (matrix_text(ctx) + "\n-1\n\n" + matrix_text(dom) + "\n" + words).encode()
        try:
            p = subprocess.run([pb.ORACLEPIP, "pip"], input=txt, capture_output=True, timeout=3)
        except subprocess.TimeoutExpired:
            continue
        if p.returncode != 0:
            continue
        g = subprocess.run([pb.REFPIP_GPU, "pip"], input=txt, capture_output=True, timeout=120)
        assert g.returncode == 0, (trial, g.stderr.decode()[-200:])
        assert pb.squash(g.stdout.decode()) == pb.squash(p.stdout.decode()), (trial, words)
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


@pytest.mark.parametrize("nvar,ni,nq,cap", [(20, 5000, 0, 8), (20, 4200, 1, 600), (200, 3000, 0, 8)])
def test_rows_beyond_lds(nvar, ni, nq, cap):
    """Tableaux whose row tables do not fit a workgroup's LDS (more than ~3,400 rows of <= 128
    columns, ~2,500 of <= 256): the engine keeps the tables in HBM (the kernel's GM instantiation),
    as the reference's expanser has no row bound (traiter.c:55-88); bit-exact vs the oracle."""
    from gpu_common import compare
    from piplib_amd import synth
    rows = synth.lexmin_batch(5, 3, nvar, ni)
    n, piv = compare(rows, nvar, 0, nq, cap_cuts=cap)
    assert n == 3 and piv > 0


@pytest.mark.parametrize("ni", [3300, 4500])
def test_tree_path_many_rows(ni):
    """The host tree (pipamd_solve_tableau) on a parametric problem with thousands of rows: ~3,300
    rows is just inside what LDS holds for <= 128 columns (the re-housing clamp of round 1 stopped
    short of it), 4,500 needs the tables in HBM; text equals the oracle's."""
    import numpy as np
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    rows = synth.lexmin_batch(9, 1, 12, ni, nnz=3, cmax=4)[0]          # 12 unknowns | constant
    rng = np.random.default_rng(ni)
    par = rng.integers(-1, 2, size=(ni, 1)).astype(np.int64)           # one parameter column
    ineq = np.concatenate([rows, par], axis=1)
    ctx = np.array([[1, 0], [-1, 6]], dtype=np.int64)                  # 0 <= p <= 6
    prob = synth.Problem(12, 1, ni, 2, -1, 1, ineq, ctx)
    o = pb.run_batch(pb.ORACLEPIP, [prob], timeout=600).results[0]
    assert o.status == pb.ST_OK
    text, piv = eng.solve_tableau(eng.Engine(0), 12, 1, ni, 2, -1, 1, ineq, ctx)
    assert pb.squash(text) == pb.squash(o.text) and piv == o.pivots


@pytest.mark.parametrize("nvar,ni,nq,waves", [(127, 64, 1, 0), (63, 32, 0, 0), (40, 30, 1, 4), (41, 30, 1, 1), (200, 90, 1, 0)])
def test_rows_fetched_by_the_pivot_kernel(nvar, ni, nq, waves):
    """PIPAMD_T_ROWS_STAY: pipamd_batch_load builds only the row tables and the first pivot launch reads the rows
    from the caller's array (bulk and four-wave launches, one and two row chunks; 41 unknowns = an even number of
    columns, 40 = odd: there the rows are copied as usual): statuses, pivot counts and solutions as without the flag,
    and as the oracle has them."""
    import torch
    from gpu_common import oracle_batch, solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    rows = synth.lexmin_batch(900 + nvar, 300 if nvar < 100 else 60, nvar, ni)
    outs = []
    for stay in (0, eng.T_ROWS_STAY):
        e = eng.Engine(0)
        if waves:
            e.set_waves_per_job(waves)
        e.set_bulk_min(64)
        b = eng.Batch(e, rows, nvar, 0, tflags=(eng.T_INT if nq else 0) | stay)
        for _ in range(2):  # the second load + solve reuses the workspace
            b.load()
            b.solve()
        b.fetch()
        torch.cuda.synchronize()
        outs.append((b.status.cpu().numpy(), b.pivots.cpu().numpy(), b.cuts.cpu().numpy(), b.sol_num.cpu().numpy(),
                     b.sol_den.cpu().numpy()))
    for x, y in zip(*outs):
        assert (x == y).all()
    st, pv, _, num, den = outs[1]
    o = oracle_batch(rows, nvar, 0, nq).results
    for k, r in enumerate(o):
        assert r.status != pb.ST_ABORT and pv[k] == r.pivots
        got = "()" if st[k] == eng.ST_NIL else pb.squash(solution_text(num[k], den[k]))
        assert got == pb.squash(r.text), k


@pytest.mark.parametrize("name,ni,nq,kw,cap,stay", [
    ("headline", 64, 1, dict(), None, True),           # most tableaux finish in the lean kernel; some leave it mid-run
    ("headline-copied-rows", 64, 1, dict(), None, False),  # rows already in the job blocks: packed in place
    ("rational", 64, 0, dict(), None, True),
    ("class-1-at-entry", 64, 1, dict(scale=40000), None, True),   # entries of 2^15 and more from the start: the mid path
    ("class-2-mid-run", 64, 1, dict(scale=1 << 24), None, True),  # ints at entry, a row beyond 2^31 after a few pivots
    ("class-2-mid-run-copied", 64, 1, dict(scale=1 << 24), None, False),
    ("beyond-32-bits", 40, 1, dict(scale=1 << 33), None, False),  # rows that cannot be packed (widened again in place)
    ("beyond-32-bits-fetched", 64, 1, dict(scale=1 << 33), None, True),
    ("overflow", 64, 1, dict(cmax=40000, x0max=3), None, True),   # "Integer overflow" (traiter.c:424,442) on every tableau
    ("class-160", 100, 1, dict(), None, True),         # the largest row-capacity class
    ("no-class", 113, 1, dict(), None, True),          # 161 row slots: beyond the largest class, the general kernel does it all
    ("spare-rows-spent", 64, 1, dict(), 6, True),      # PIPAMD_ST_CAPACITY inside the lean kernel, then expanser
])
def test_lean_kernel_paths(name, ni, nq, kw, cap, stay):
    """The lean bulk kernel (csrc/pip_lean.h: 127 unknowns, int rows; entries below 2^15: 24-bit products, below 2^31:
    64-bit products on the same int rows) and every way a tableau leaves it -- finished, pivot budget, a row that no longer
    fits ints (stored in the general format mid-pivot), no spare row, rows it cannot pack -- against the same batch
    without it (pipamd_debug_lean: statuses, pivot and cut counts, solutions identical) and against the oracle."""
    import torch
    from gpu_common import oracle_batch, solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    nvar, batch = 127, 160
    kw = dict(kw)
    scale = kw.pop("scale", 0)
    # (seeds on which Gomory's cuts converge: 4121 holds a tableau the reference itself does not finish within minutes)
    rows = synth.lexmin_batch({"no-class": 4112}.get(name, 4000 + ni + len(name)), batch, nvar, ni, **kw)
    if scale:  # every other tableau: one inequality multiplied through (same polyhedron, large entries)
        rows[::2, ni // 2, :] *= scale
    outs, launches = [], []
    for lean in (0, 1):
        e = eng.Engine(0)
        e.set_bulk_min(64)
        e.set_max_rows(ni + 1024)
        e.debug_lean(lean)
        b = eng.Batch(e, rows, nvar, 0, tflags=(eng.T_INT if nq else 0) | (eng.T_ROWS_STAY if stay else 0), cap_cuts=cap)
        for _ in range(2):  # the second load + solve reuses the workspace
            b.load()
            b.solve()
        launches.append(e.last_solve_launches())
        b.fetch()
        torch.cuda.synchronize()
        outs.append((b.status.cpu().numpy(), b.pivots.cpu().numpy(), b.cuts.cpu().numpy(), b.sol_num.cpu().numpy(),
                     b.sol_den.cpu().numpy()))
    if name != "no-class":
        assert launches[1] > launches[0] or cap, launches  # the lean launch went out (one more launch than without)
        # the lean launch on its own: how its tableaux ended (PipJob of csrc/pip_job.h, 200 bytes: status at byte 72,
        # pivots at 80, the largest magnitude class among its rows at 160, the lean kernel's exit reason at 172 -- 1 pivot
        # budget, 2 a row beyond ints)
        e.debug_single_launch(2)
        b.load()
        b.solve()
        e.debug_single_launch(0)
        j = b.ws[:25 * batch].view(torch.int32).view(batch, 50).cpu().numpy()
        status, npiv, why, mcls = j[:, 18], j[:, 20], j[:, 43], j[:, 40]
        running = status == eng.ST_RUN
        if name == "class-160":   # long tableaux: most spend the launch's pivot budget
            assert (npiv > 0).all() and (running & (why == 1)).any()
        if name in ("headline", "headline-copied-rows", "rational"):
            # (rows between 2^15 and 2^31 do not end a lean run any more: only the pivot budget does on this family)
            assert (~running).sum() > batch // 2 and (why[running] == 1).all(), (running.sum(), why[running])
        if name == "class-1-at-entry":   # the scaled tableaux run on the mid path from their first pivot
            assert (npiv > 0).all() and (~running[::2]).sum() > batch // 4 and (why[::2] != 2).sum() > batch // 4, why[::2]
        if name.startswith("class-2-mid-run"):   # scaled tableaux start as int rows and leave mid-run on a row beyond 2^31
            left = running[::2] & (why[::2] == 2)
            assert (npiv > 0).all() and left.sum() > batch // 8 and (mcls[::2][left] >= 2).all(), (left.sum(), why[::2])
        if name.startswith("beyond-32-bits"):   # not taken: header untouched
            assert (running[::2] & (npiv[::2] == 0)).all() and (npiv[1::2] > 0).all()
        if name == "spare-rows-spent":
            assert (status == eng.ST_CAPACITY).any()
    else:
        assert launches[1] == launches[0], launches
    for x, y in zip(*outs):
        assert (x == y).all()
    st, pv, _, num, den = outs[1]
    o = oracle_batch(rows, nvar, 0, nq).results
    for k, r in enumerate(o):
        if r.status == pb.ST_ABORT:
            assert st[k] == {2: eng.ST_OVERFLOW, 4: eng.ST_MAXCOL}.get(r.abort_code, eng.ST_OVERFLOW), (k, st[k], r.abort_code)
            continue
        assert st[k] in (eng.ST_SOLUTION, eng.ST_NIL), (k, st[k])
        assert pv[k] == r.pivots, (k, pv[k], r.pivots)
        got = "()" if st[k] == eng.ST_NIL else pb.squash(solution_text(num[k], den[k]))
        assert got == pb.squash(r.text), k


@pytest.mark.parametrize("nvar,ni,nq,stay", [
    (63, 32, 0, True),     # BASELINE configs[1]: 32 x 64, rational
    (63, 32, 1, True),
    (41, 30, 1, True),     # 42 columns (even): the rows are fetched by the kernel
    (40, 30, 1, True),     # 41 columns (odd): pipamd_batch_load copies them, the kernel packs them in place
    (100, 60, 1, False),
    (5, 8, 1, True),
    (126, 64, 1, True),    # 127 columns -> rows of 128 with a zero column
])
def test_lean_kernel_other_widths(nvar, ni, nq, stay):
    """The lean bulk kernel on tableaux of fewer than 127 unknowns (its instantiation with run-time column counts):
    identical to the same batch without it and to the oracle."""
    import torch
    from gpu_common import oracle_batch, solution_text
    import pipbatch as pb
    from piplib_amd import engine as eng, synth
    batch = 200
    kw = dict(nnz=3, cmax=3, x0max=5) if nvar < 10 else {}
    rows = synth.lexmin_batch(5000 + nvar, batch, nvar, ni, **kw)
    outs, launches = [], []
    for lean in (0, 1):
        e = eng.Engine(0)
        e.set_bulk_min(64)
        e.set_max_rows(ni + 1024)
        e.debug_lean(lean)
        b = eng.Batch(e, rows, nvar, 0, tflags=(eng.T_INT if nq else 0) | (eng.T_ROWS_STAY if stay else 0))
        for _ in range(2):
            b.load()
            b.solve()
        launches.append(e.last_solve_launches())
        b.fetch()
        torch.cuda.synchronize()
        outs.append((b.status.cpu().numpy(), b.pivots.cpu().numpy(), b.cuts.cpu().numpy(), b.sol_num.cpu().numpy(),
                     b.sol_den.cpu().numpy()))
    assert launches[1] > launches[0], launches
    for x, y in zip(*outs):
        assert (x == y).all()
    st, pv, _, num, den = outs[1]
    o = oracle_batch(rows, nvar, 0, nq).results
    for k, r in enumerate(o):
        assert r.status != pb.ST_ABORT and st[k] in (eng.ST_SOLUTION, eng.ST_NIL), (k, st[k])
        assert pv[k] == r.pivots, (k, pv[k], r.pivots)
        got = "()" if st[k] == eng.ST_NIL else pb.squash(solution_text(num[k], den[k]))
        assert got == pb.squash(r.text), k


def test_lone_batches_option():
    """pipamd_engine_set_lone_batches: the tableaux the lean launch leaves go straight to the tail launches (one launch
    less); statuses, pivot and cut counts and solutions are those of the default sequence."""
    import torch
    from piplib_amd import engine as eng, synth
    nvar, ni = 127, 64
    rows = synth.lexmin_batch(6100, 600, nvar, ni)
    outs, launches = [], []
    for lone in (0, 1):
        e = eng.Engine(0)
        e.set_bulk_min(64)
        e.set_max_rows(ni + 1024)
        e.set_lone_batches(lone)
        b = eng.Batch(e, rows, nvar, 0, tflags=eng.T_INT | eng.T_ROWS_STAY)
        b.load()
        b.solve()
        launches.append(e.last_solve_launches())
        b.fetch()
        torch.cuda.synchronize()
        outs.append((b.status.cpu().numpy(), b.pivots.cpu().numpy(), b.cuts.cpu().numpy(), b.sol_num.cpu().numpy(),
                     b.sol_den.cpu().numpy()))
    assert launches[1] < launches[0], launches
    for x, y in zip(*outs):
        assert (x == y).all()
    assert (outs[0][0] != eng.ST_RUN).all()
