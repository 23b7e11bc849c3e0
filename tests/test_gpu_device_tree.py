"""-m gpu: the device-resident traiter() (piplib_amd/csrc/pip_quast.hip) behind
pipamd_solve_tableaux_lockstep.  One wave per problem runs pivots, compa_test sub-problems, quast
forks and parametric cuts on the GPU; the checks here:

* random parametric / rational / big-parameter problems against the CPU oracle (tape text and
  pivot count of every problem), with the share the device tree really served asserted;
* the reference's own .dat goldens through the device tree, problem by problem;
* problems the kernel must hand back (64-bit overflow, too many rows) still come out right through
  the fallback, and with the device tree switched off the answers are the same."""
import os
import subprocess

import numpy as np
import pytest

import pipbatch as pb
from datfile import read_dat
from test_oracle_golden import PIPTEST_DAT

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
G = os.path.join(pb.ROOT, "tests", "golden")


def _screen(probs, limit=3000, flags=0, exe=None):
    """(problem, oracle result) for the problems the CPU oracle (exe: the 64-bit one, or pb.ORACLEPIP128) finishes quickly"""
    keep = []
    for p in probs:
        try:
            r = pb.run_batch(exe or pb.ORACLEPIP, [p], flags, timeout=3).results[0]
        except subprocess.TimeoutExpired:
            continue
        if r.pivots <= limit:
            keep.append((p, r))
    return keep


def _check(keep, min_served, deepest=False):
    from piplib_amd import engine as eng
    e = eng.Engine(0)
    probs = [p for p, _ in keep]
    got = eng.solve_tableaux(e, probs, lockstep=True, deepest_cut=deepest)
    served, back = e.last_device_tree()
    assert served + back <= len(probs)
    assert served >= min_served * len(probs), (served, back, len(probs))
    for (p, r), (text, rc, st, piv) in zip(keep, got):
        if r.status == pb.ST_ABORT:
            assert rc == -5, (rc, st)
            continue
        assert rc == 0, (rc, st)
        want = "void\n" if r.status == pb.ST_VOID else r.text
        assert pb.squash(text) == pb.squash(want)
        assert piv == r.pivots
    e.set_device_tree(False)  # the forest / tree alone: same answers, entry by entry
    assert eng.solve_tableaux(e, probs, lockstep=True, deepest_cut=deepest) == got
    assert e.last_device_tree() == (0, 0)
    return served, back


@pytest.mark.parametrize("seed,shape,nq,count,served,cmax", [
    (141, (5, 2, 7, 2), 1, 120, 0.9, 4), (142, (4, 3, 6, 3), 1, 120, 0.9, 4), (143, (6, 1, 8, 1), 1, 120, 0.9, 4),
    (144, (16, 3, 20, 3), 1, 60, 0.6, 4), (145, (8, 2, 10, 2), 0, 120, 0.9, 4), (146, (6, 0, 9, 0), 1, 120, 0.9, 4),
    (147, (10, 4, 14, 1), 1, 60, 0.6, 4), (148, (3, 5, 6, 4), 1, 120, 0.8, 4),
    # more than 64 logical rows (two passes of every lane-per-row loop) and tableaux 60+ columns wide; entries of
    # magnitude 1 (with 4 three quarters of such problems leave 64 bits and are handed back)
    (149, (30, 2, 45, 2), 0, 40, 0.5, 1), (150, (50, 3, 40, 2), 0, 40, 0.3, 1), (157, (28, 1, 44, 1), 1, 30, 0.3, 1)])
def test_random_problems_vs_oracle(seed, shape, nq, count, served, cmax):
    """integer and rational solves, with and without parameters / context rows.  (Unscreened random
    16x20 problems run to hundreds of cuts and new parameters; a quarter of them overflow 64 bits --
    the reference wraps -- and are handed back, hence the lower share asked of the device tree there.)"""
    from piplib_amd import synth
    keep = _screen(synth.random_problems(seed, count, *shape, nq, cmax=cmax))
    assert len(keep) >= 0.4 * count  # (the 10x14 family has many problems the reference itself does not finish)
    _check(keep, served)


@pytest.mark.parametrize("seed,shape,nq,count,served,cmax", [
    (144, (16, 3, 20, 3), 1, 60, 0.9, 4), (147, (10, 4, 14, 1), 1, 60, 0.9, 4), (142, (4, 3, 6, 3), 1, 120, 0.9, 4),
    (149, (30, 2, 45, 2), 0, 40, 0.9, 1), (157, (28, 1, 44, 1), 1, 30, 0.9, 1)])
def test_random_problems_128bit_device_tree_vs_oracle128(seed, shape, nq, count, served, cmax):
    """The device-resident traiter() of the overflow-safe flavour (the kernel is a template over the entry type: 128-bit
    tableaux, contexts, cuts and tape cells in LDS, every product and sum checked against 128 bits): the families on
    which the 64-bit kernel hands a quarter to three quarters of the problems back for overflow, through
    pipamd_solve_tableaux_lockstep128, against the 128-bit CPU oracle -- tape text and pivot count of every problem, the
    share served on the device asserted -- and again with the device tree off (ForestT<__int128> / TreeT<__int128>)."""
    from piplib_amd import engine as eng, synth
    keep = _screen(synth.random_problems(seed, count, *shape, nq, cmax=cmax), exe=pb.ORACLEPIP128)
    assert len(keep) >= 0.4 * count
    e = eng.Engine(0)
    probs = [p for p, _ in keep]
    got = eng.solve_tableaux_lockstep128(e, probs)
    srv, back = e.last_device_tree()
    assert srv + back <= len(probs) and srv >= served * len(probs), (srv, back, len(probs))
    for (p, r), (text, rc, st, piv) in zip(keep, got):
        if r.status == pb.ST_ABORT:
            assert rc == -5, (rc, st)
            continue
        assert rc == 0, (rc, st)
        assert pb.squash(text) == pb.squash("void\n" if r.status == pb.ST_VOID else r.text)
        assert piv == r.pivots
    e.set_device_tree(False)
    assert eng.solve_tableaux_lockstep128(e, probs) == got
    assert e.last_device_tree() == (0, 0)


def test_big_parameter_problems_vs_oracle():
    """a big parameter (traiter.c:106-121, integrer.c:382-385): every parameter column in turn"""
    from piplib_amd import synth
    probs = []
    for k, p in enumerate(synth.random_problems(151, 150, 5, 3, 7, 2, 1)):
        probs.append(pb.Problem(p.nvar, p.nparm, p.ni, p.nc, p.nvar + 1 + k % p.nparm, p.nq, p.ineq, p.ctx))
    keep = _screen(probs)
    assert len(keep) >= 100
    _check(keep, 0.9)


def test_deepest_cut_vs_oracle():
    """the deepest-cut option (integrer.c:417-438) runs on the device too"""
    from piplib_amd import synth
    probs = synth.random_problems(152, 100, 5, 2, 7, 2, 1) + synth.random_problems(153, 100, 6, 0, 9, 0, 1)
    keep = _screen(probs, flags=pb.F_DEEPEST)
    assert len(keep) >= 120
    _check(keep, 0.9, deepest=True)


def test_unsimplified_inputs_vs_oracle():
    from piplib_amd import engine as eng, synth
    keep = _screen(synth.random_problems(154, 100, 5, 2, 7, 2, 1, cmax=9), flags=pb.F_NOSIMPLIFY)
    e = eng.Engine(0)
    got = eng.solve_tableaux(e, [p for p, _ in keep], lockstep=True, simplify=False)
    assert e.last_device_tree()[0] >= 0.9 * len(keep)
    for (p, r), (text, rc, st, piv) in zip(keep, got):
        if r.status == pb.ST_ABORT:
            assert rc == -5
            continue
        assert rc == 0 and piv == r.pivots
        assert pb.squash(text) == pb.squash("void\n" if r.status == pb.ST_VOID else r.text)


def _quotient_family():
    """problems in which several unknowns need the same quotient floor(p / D): the second cut must find the
    parameter the first one declared (find_parm, integrer.c:258-291) instead of declaring it again (a build with
    a counter saw 70 such finds over the family, and none in the random families above)"""
    def prob(rows, ctx, nvar, nparm):
        rows = np.array(rows, dtype=np.int64)
        ctx = np.array(ctx, dtype=np.int64).reshape(-1, nparm + 1)
        return pb.Problem(nvar, nparm, rows.shape[0], ctx.shape[0], -1, 1, rows, ctx)
    out = []
    for D in (2, 3, 4, 5, 7):
        for shift in (0, 1, 2):
            # D x1 >= p + shift, D x2 >= p + shift [, D x3 >= p + shift]
            out.append(prob([[D, 0, -shift, -1], [0, D, -shift, -1]], [[1, 0]], 2, 1))
            out.append(prob([[D, 0, 0, -shift, -1], [0, D, 0, -shift, -1], [0, 0, D, -shift, -1]], [[1, 0]], 3, 1))
        # two parameters, one quotient shared by two unknowns and another one on its own
        out.append(prob([[D, 0, 0, 0, -1, 0], [0, D, 0, 0, -1, 0], [0, 0, 3, 0, 0, -1], [-1, 0, 3, 1, 0, -1]],
                        [[1, 0, 0], [0, 1, 0]], 3, 2))
        out.append(prob([[D, 0, 0, -1, -1], [0, D, 0, -1, -1], [1, 1, -3, 0, 0]], [[1, 0, 0], [0, 1, 0]], 2, 2))
    return out


def test_quotient_parameter_found_again():
    keep = _screen(_quotient_family())
    assert len(keep) >= 35
    # the family does what it is for: fewer parameters declared than rows cut in at least most problems
    assert sum(r.text.count("newparm") == 1 for _, r in keep) >= 10
    served, back = _check(keep, 1.0)
    assert back == 0


@pytest.mark.timeout(180)
def test_overflow_in_the_pivot_column_tournament_terminates():
    """found by tests/manual/fuzz_param.py (FUZZ_BIG): products that overflow in choisir_piv's cross-multiplication
    compare as garbage and sent the tournament round in circles for ever; now the problem is handed back"""
    from piplib_amd import synth
    keep = _screen(synth.random_problems(998034085, 8, 34, 6, 45, 3, 1, cmax=2, bmax=12))
    assert len(keep) >= 2
    served, back = _check(keep, 0.0)
    assert back >= 1


def test_overflowing_problems_are_handed_back():
    """coefficients that overflow 64 bits: the device tree must notice (it computes on true integers)
    and the fallback reproduces the reference's wrap-around / "Integer overflow" behaviour"""
    from piplib_amd import synth
    keep = _screen(synth.random_problems(155, 60, 6, 2, 9, 2, 1, cmax=3000000, bmax=2000000000), limit=20000)
    assert len(keep) >= 30
    served, back = _check(keep, 0.0)
    assert back >= 5, (served, back)   # the family really overflows


@pytest.mark.parametrize("seed,shape,nq,count,served", [
    (156, (6, 1, 70, 1), 1, 12, 0.9), (158, (10, 2, 90, 2), 1, 16, 0.8), (159, (8, 0, 100, 0), 1, 16, 0.9), (160, (12, 3, 64, 2), 0, 16, 0.9)])
def test_tall_problems_on_the_device(seed, shape, nq, count, served):
    """57 ... 104 inequalities (round 4: up to 128 real rows; tab_sort_rows with two rows per lane, sort_rows_tall): served
    on the device, tape and pivot count as the oracle has them; with the device tree off the same answers"""
    from piplib_amd import synth
    keep = _screen(synth.random_problems(seed, count, *shape, nq))
    assert len(keep) >= count // 2
    _check(keep, served)


def test_taller_problems_are_handed_back():
    """more than 104 inequalities: the shape test sends them to the forest"""
    from piplib_amd import engine as eng, synth
    keep = _screen(synth.random_problems(161, 8, 6, 1, 110, 1, 1))
    e = eng.Engine(0)
    got = eng.solve_tableaux(e, [p for p, _ in keep], lockstep=True)
    assert e.last_device_tree() == (0, 0)
    for (p, r), (text, rc, st, piv) in zip(keep, got):
        if r.status != pb.ST_ABORT:
            assert rc == 0 and piv == r.pivots and pb.squash(text) == pb.squash("void\n" if r.status == pb.ST_VOID else r.text)


@pytest.mark.parametrize("name", PIPTEST_DAT)
def test_dat_goldens_through_the_device_tree(name):
    """test/*.dat of the reference (test/Makefile.am PIPTEST) against its .ll files"""
    from piplib_amd import engine as eng
    e = eng.Engine(0)
    dat = read_dat(os.path.join(G, "test", name + ".dat"))
    probs = [pb.Problem(p["nvar"], p["nparm"], p["ni"], p["nc"], p["bigparm"], p["nq"],
                        np.asarray(p["ineq"], dtype=np.int64).reshape(p["ni"], p["nvar"] + p["nparm"] + 1),
                        np.asarray(p["ctx"], dtype=np.int64).reshape(p["nc"], p["nparm"] + 1)) for p in dat]
    got = eng.solve_tableaux(e, probs, lockstep=True)
    out = []
    for p, (text, rc, st, piv) in zip(dat, got):
        assert rc == 0, (rc, st)
        out.append("(" + p["comment"] + (text if text == "void\n" else ")\n" + text) + ")\n")
    want = open(os.path.join(G, "test", name + ".ll"), encoding="latin-1").read()
    assert pb.squash("".join(out)) == pb.squash(want)


def test_goldens_mostly_run_on_the_device():
    """of the reference's .dat problems that fit the device tree's shape limits, nearly all are served there"""
    from piplib_amd import engine as eng
    e = eng.Engine(0)
    probs = []
    for name in PIPTEST_DAT:
        for p in read_dat(os.path.join(G, "test", name + ".dat")):
            probs.append(pb.Problem(p["nvar"], p["nparm"], p["ni"], p["nc"], p["bigparm"], p["nq"],
                                    np.asarray(p["ineq"], dtype=np.int64).reshape(p["ni"], p["nvar"] + p["nparm"] + 1),
                                    np.asarray(p["ctx"], dtype=np.int64).reshape(p["nc"], p["nparm"] + 1)))
    eng.solve_tableaux(e, probs, lockstep=True)
    served, back = e.last_device_tree()
    assert served >= 0.8 * len(probs), (served, back, len(probs))


def test_mixed_shapes_in_one_call():
    """problems of very different sizes in one call: launches go by size class, answers stay in input order"""
    from piplib_amd import synth
    probs = []
    for seed, shape, count, cmax in ((161, (3, 1, 4, 1), 60, 4), (162, (16, 3, 20, 3), 12, 4), (163, (30, 2, 45, 2), 6, 1),
                                     (164, (5, 2, 7, 2), 60, 4)):
        probs += synth.random_problems(seed, count, *shape, 1, cmax=cmax)
    rng = np.random.default_rng(7)
    probs = [probs[i] for i in rng.permutation(len(probs))]
    keep = _screen(probs)
    assert len(keep) >= 100
    _check(keep, 0.7)
