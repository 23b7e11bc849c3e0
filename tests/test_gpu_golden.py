"""-m gpu: PipLib's own test inputs against the reference's golden .ll files, two ways:
(1) through pipamd_solve_tableau (layer 3 of the C ABI: host decision tree + HIP engine; the tape
    cells are printed by the Python harness), and
(2) through the reference's own front ends -- its parsers, pip_solve, solution tape, sol_edit and
    pip_quast_print -- with only traiter() bound to the engine (bindings/piplib_traiter_hook.c,
    built as oracle/_ref/refpip_gpu by oracle/Makefile)."""
import json
import os

import pytest

import pipbatch as pb
from datfile import read_dat
from test_oracle_golden import PIPTEST_DAT, MANIFEST

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
G = os.path.join(pb.ROOT, "tests", "golden")


def solve_file(path, bits=64, device_tree=True, **kw):
    from piplib_amd import engine as eng
    e = eng.Engine(0)
    e.set_device_tree(device_tree)  # False: every problem through the host decision tree
    out = []
    status = None
    for p in read_dat(path):
        out.append("(" + p["comment"])
        try:
            text, _ = eng.solve_tableau(e, p["nvar"], p["nparm"], p["ni"], p["nc"], p["bigparm"], p["nq"],
                                        p["ineq"], p["ctx"], bits=bits, **kw)
        except eng.SolverError as ex:
            status = ex.status
            break
        out.append(text if text == "void\n" else ")\n" + text)
        out.append(")\n")
    return "".join(out), status


@pytest.mark.parametrize("device_tree", [True, False], ids=["device-tree", "host-tree"])
@pytest.mark.parametrize("name", PIPTEST_DAT)
def test_dat_golden_on_gpu(name, device_tree):
    got, status = solve_file(os.path.join(G, "test", name + ".dat"), device_tree=device_tree)
    assert status is None
    want = open(os.path.join(G, "test", name + ".ll"), encoding="latin-1").read()
    assert pb.squash(got) == pb.squash(want)


@pytest.mark.parametrize("device_tree", [True, False], ids=["device-tree", "host-tree"])
@pytest.mark.parametrize("key", sorted(MANIFEST))
def test_ref_generated_on_gpu(key, device_tree):
    m = MANIFEST[key]
    got, status = solve_file(os.path.join(G, key), device_tree=device_tree)
    want = open(os.path.join(G, "ref_dp", m["ll"]), encoding="latin-1").read()
    if m["rc"] != 0:  # the reference exit(1)s with "Integer overflow": same verdict, same partial output
        from piplib_amd import engine as eng
        assert status == eng.ST_OVERFLOW
    else:
        assert status is None
    assert pb.squash(got) == pb.squash(want)


from test_oracle_golden import PIPTEST_PIP  # noqa: E402

needs_hook = pytest.mark.skipif(not pb.have_ref_gpu(), reason="oracle/_ref/refpip_gpu not built (no /root/reference)")


def ref_front_end(args, stdin=None, timeout=300, device_tree=True):
    """oracle/_ref/refpip_gpu: the reference's front end; every traiter() call goes to the GPU
    (small problems to the device-resident traiter() unless PIPAMD_NO_DEVICE_TREE is set)."""
    import subprocess
    env = dict(os.environ)
    if not device_tree:
        env["PIPAMD_NO_DEVICE_TREE"] = "1"
    return subprocess.run([pb.REFPIP_GPU] + args, stdin=stdin, capture_output=True, timeout=timeout, env=env)


@needs_hook
@pytest.mark.parametrize("device_tree", [True, False], ids=["device-tree", "host-tree"])
@pytest.mark.parametrize("name", PIPTEST_DAT)
def test_dat_golden_through_reference_front_end(name, device_tree):
    """test/*.dat: the reference's tab_get, tape and sol_edit around the GPU traiter vs test/*.ll."""
    p = ref_front_end(["dat", os.path.join(G, "test", name + ".dat")], device_tree=device_tree)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    want = open(os.path.join(G, "test", name + ".ll"), encoding="latin-1").read()
    assert pb.squash(p.stdout.decode("latin-1")) == pb.squash(want)


@needs_hook
@pytest.mark.parametrize("key", sorted(MANIFEST))
def test_ref_generated_through_reference_front_end(key):
    """inputs without a usable .ll in the reference tree (both "Integer overflow" inputs among them)"""
    m = MANIFEST[key]
    p = ref_front_end(["dat"] + m.get("args", []) + [os.path.join(G, key)])
    want = open(os.path.join(G, "ref_dp", m["ll"]), encoding="latin-1").read()
    if m["rc"] != 0:
        assert p.returncode != 0 and m["stderr"] in p.stderr.decode()
    else:
        assert p.returncode == 0, p.stderr.decode()[-300:]
    assert pb.squash(p.stdout.decode("latin-1")) == pb.squash(want)


@needs_hook
@pytest.mark.parametrize("name", PIPTEST_PIP)
def test_pip_solve_golden_on_gpu(name):
    """example/*.pip through the reference's pip_solve (piplib.c:722-880, unchanged) with its two
    traiter calls bound to the engine; printed by the reference's pip_quast_print; vs example/*.ll."""
    with open(os.path.join(G, "example", name + ".pip")) as f:
        p = ref_front_end(["pip"], stdin=f)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    want = open(os.path.join(G, "example", name + ".ll"), encoding="latin-1").read()
    assert pb.squash(p.stdout.decode("latin-1")) == pb.squash(want)


@needs_hook
@pytest.mark.parametrize("device_tree", [True, False], ids=["device-tree", "host-tree"])
@pytest.mark.parametrize("name", ["small", "square", "max", "big", "cg1", "sven"])
def test_compute_dual_on_gpu(name, device_tree):
    """pip_solve with Nq = 0 and Compute_dual = 1 (TRAITER_DUAL through the hook) vs reference-generated fixtures; with
    the device-resident traiter() (which computes the dual since round 3) and with the host tree alone."""
    d = os.path.join(G, "ref_dp")
    with open(os.path.join(d, f"dual__{name}.pip")) as f:
        p = ref_front_end(["pip"], stdin=f, device_tree=device_tree)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    want = open(os.path.join(d, f"dual__{name}.ll"), encoding="latin-1").read()
    assert pb.squash(p.stdout.decode("latin-1")) == pb.squash(want)


def test_plain_c_example():
    """examples/solve_small.c (built by __graft_entry__.build()): the C ABI from plain C -- a
    traiter() call and a maind.c-style call; the cells it prints, formatted as sol_edit would,
    equal the oracle's text."""
    import subprocess
    import numpy as np
    from piplib_amd import engine as eng, synth
    exe = os.path.join(pb.ROOT, "examples", "solve_small")
    if not os.access(exe, os.X_OK):
        import __graft_entry__
        __graft_entry__.build()
    p = subprocess.run([exe], capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    tapes, cur = [], None
    for ln in p.stdout.decode().splitlines():
        if ln.startswith("tape"):
            cur = []
            tapes.append(cur)
        elif ln.startswith("cell"):
            cur.append(tuple(int(x) for x in ln.split()[1:4]))
    assert len(tapes) == 2
    # 1. lexmin of (i, j): i - 3j + 12 >= 0, -2i + j + 3 >= 0 (the reference's example/small.pip as a tableau)
    small = synth.Problem(2, 0, 2, 0, -1, 1, np.array([[1, -3, 12], [-2, 1, 3]], dtype=np.int64), np.zeros((0, 1), np.int64))
    # 2. one parametric problem with a context row
    prob = synth.Problem(2, 1, 3, 1, -1, 1, np.array([[1, 1, 0, -1], [-1, 0, 5, 0], [0, -1, 7, 0]], dtype=np.int64),
                         np.array([[-1, 12]], dtype=np.int64))
    o = pb.run_batch(pb.ORACLEPIP, [small, prob])
    assert pb.squash(eng.tape_text(tapes[0])) == pb.squash(o.results[0].text)
    assert pb.squash(eng.tape_text(tapes[1])) == pb.squash(o.results[1].text)


@pytest.mark.parametrize("name,dom,ctx,opts", [
    # found by tests/manual/fuzz_pipsolve.py: (1) a sub-problem of compa_test behind the first negative
    # row never terminates (the reference never looks at it); (2) 179,996 pivots and > 768 rows
    ("speculative_compa",
     [[0, -3, 2, 0, 1, -3, -3, 1, -4], [1, -1, -2, -3, -1, 3, -2, -2, 7], [1, 1, 2, -2, 3, 1, 3, 0, 9],
      [0, 0, -1, -2, 3, 1, 3, 1, -2], [1, 2, 3, -1, 2, -3, 1, 0, 1], [1, -1, 2, -2, -1, -2, 1, -1, 8],
      [1, 0, -1, -1, -3, 0, 0, 2, 4]],
     [[1, -2, -1, 3], [1, 0, 1, 1]], {"Urs_parms": 1}),
    ("many_rows",
     [[1, 3, 2, 2, -3, 1, 1, 0, 1], [1, 3, 1, -2, -2, -2, -2, 0, 7], [1, -1, 2, 3, 2, 0, -3, -3, -1],
      [1, -1, -2, -3, 2, 0, 1, 1, 10], [0, -2, 3, 2, -1, -1, 1, 1, 1]],
     [[1, 2, -1, -2, 6], [1, -2, 1, 0, 0]], {}),
    ("sub_problem_2000_cuts",
     [[1, -2, 0, 1, 3, -3, 0, -1, -1], [1, -1, 1, 3, 0, 0, 1, -2, 10], [1, -1, 0, 3, -3, 2, -1, 0, -2],
      [0, 2, -2, 1, -2, 2, -3, -3, 2], [0, -2, 3, -3, -2, 0, 2, -2, -4], [1, 0, 2, -2, 1, -2, 3, 0, -6],
      [0, 3, 1, 2, -1, -1, 3, 3, 5], [1, 3, 1, 3, -2, 2, -1, 2, 6], [1, -1, -3, -3, -1, -3, -3, 0, -5]],
     [[1, 2, -2, 0, 1], [1, 2, -1, 0, 4]], {"Urs_parms": 1}),
])
@needs_hook
def test_pip_solve_fuzz_regressions(name, dom, ctx, opts):
    """The reference's pip_solve over the GPU traiter vs the oracle's `pip` mode on inputs that once
    failed with PIPAMD_ST_CAPACITY."""
    import subprocess
    import numpy as np
    from datfile import matrix_text
    dom, ctx = np.array(dom, dtype=np.int64), np.array(ctx, dtype=np.int64)
    words = "".join(k + "\n" for k in opts)
    txt = (matrix_text(ctx) + "\n-1\n\n" + matrix_text(dom) + "\n" + words).encode()
    o = subprocess.run([pb.ORACLEPIP, "pip"], input=txt, capture_output=True, timeout=120)
    assert o.returncode == 0
    g = subprocess.run([pb.REFPIP_GPU, "pip"], input=txt, capture_output=True, timeout=600)
    assert g.returncode == 0, g.stderr.decode()[-300:]
    assert pb.squash(g.stdout.decode("latin-1")) == pb.squash(o.stdout.decode("latin-1"))


# ---------------------------------------------------------------- 128-bit host tree (layer 3)
@pytest.mark.parametrize("name", PIPTEST_DAT)
def test_dat_golden_on_gpu_128bit_tree(name):
    """The reference's .dat suite through pipamd_solve_tableau128: 128-bit entries on the device, in
    the context, the parametric cuts and the tape.  Nothing overflows on these inputs, so the
    overflow-safe flavour must print the reference's int64 goldens."""
    got, status = solve_file(os.path.join(G, "test", name + ".dat"), bits=128)
    assert status is None
    want = open(os.path.join(G, "test", name + ".ll"), encoding="latin-1").read()
    assert pb.squash(got) == pb.squash(want)


@pytest.mark.parametrize("seed", [81, 82, 83, 84])
def test_parametric_128bit_tree_vs_reference_gmp_build(seed):
    """Parametric problems with coefficients large enough that the int64 build overflows on part of them:
    pipamd_solve_tableau128 (TreeT<__int128>: device tableaux, context, parametric cuts and tape in 128 bits) against
    what the REFERENCE's own GMP build prints for them (tests/golden/gmp/param<seed>.json): same quast text, same
    number of pivoter calls, on every problem on which no value leaves the signed 128-bit range."""
    from gmpfix import gmp_fixture
    from piplib_amd import engine as eng
    probs, flags, recs, sha = gmp_fixture("param%d" % seed)
    e = eng.Engine(0)
    checked = 0
    for i, (p, r) in enumerate(zip(probs, recs)):
        if "status" not in r or r["wrap128"] or r["pivots"] > 20000:
            continue   # the reference did not finish in time / a fixed-width run may wrap / too long for a test
        try:
            text, piv = eng.solve_tableau(e, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, p.ineq, p.ctx, bits=128)
        except eng.SolverError as ex:
            if r["status"] != pb.ST_ABORT:
                # only the limb-determinant "Integer overflow" of the fixed-width flavours may differ from the GMP build
                o128 = pb.run_batch(pb.ORACLEPIP128, [p], timeout=60).results[0]
                assert ex.status == eng.ST_OVERFLOW and o128.status == pb.ST_ABORT and o128.abort_code == 2, (seed, i)
            continue
        assert r["status"] != pb.ST_ABORT, (seed, i)
        want = "void\n" if r["status"] == pb.ST_VOID else r["text"]
        assert pb.squash(text) == pb.squash(want) and piv == r["pivots"], (seed, i, piv, r["pivots"])
        checked += 1
    assert checked >= 12


def test_many_problems_on_the_128bit_tree():
    """pipamd_solve_tableaux128 (host threads, a TreeT<__int128> each) gives problem by problem what the one-problem
    entry gives -- which the test above holds against the reference's GMP build."""
    from gmpfix import gmp_fixture
    from piplib_amd import engine as eng
    probs, flags, recs, sha = gmp_fixture("param81")
    keep = [p for p, r in zip(probs, recs) if "status" in r and r["pivots"] <= 20000]
    e = eng.Engine(0)
    many = eng.solve_tableaux128(e, keep, nthreads=6)
    assert len(many) == len(keep) >= 20
    for p, (text, rc, st, piv) in zip(keep, many):
        try:
            t1, p1 = eng.solve_tableau(e, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, p.ineq, p.ctx, bits=128)
        except eng.SolverError as ex:
            assert rc == -5 and st == ex.status
            continue
        assert rc == 0 and text == t1 and piv == p1


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [81, 82, 83, 84])
def test_parametric_128bit_lockstep_vs_reference_gmp_build(seed):
    """The same families through pipamd_solve_tableaux_lockstep128 -- ForestT<__int128>: one clone / patch / pivot-kernel /
    gather sequence per step for all problems, device tableaux, contexts, parametric cuts and tape cells 128-bit --
    against the reference's GMP build (quast text and number of pivoter calls), and problem by problem against the
    one-problem entry."""
    from gmpfix import gmp_fixture
    from piplib_amd import engine as eng
    probs, flags, recs, sha = gmp_fixture("param%d" % seed)
    keep = [(p, r) for p, r in zip(probs, recs) if "status" in r and not r["wrap128"] and r["pivots"] <= 20000]
    e = eng.Engine(0)
    many = eng.solve_tableaux_lockstep128(e, [p for p, _ in keep])
    assert len(many) == len(keep) >= 12
    # the device-resident traiter() of the 128-bit flavour serves them (one wave per problem, no host round trip) ...
    served, back = e.last_device_tree()
    assert served >= 0.95 * len(keep), (served, back, len(keep))
    # ... and without it the lock-step scheduler gives the same, entry by entry
    e.set_device_tree(False)
    assert eng.solve_tableaux_lockstep128(e, [p for p, _ in keep]) == many and e.last_device_tree() == (0, 0)
    e.set_device_tree(True)
    checked = 0
    for i, ((p, r), (text, rc, st, piv)) in enumerate(zip(keep, many)):
        if rc != 0:
            # "Integer overflow" of the limb determinant (fixed-width flavours only) or a problem the reference aborts on:
            # the one-problem entry must say the same
            with pytest.raises(eng.SolverError) as ex:
                eng.solve_tableau(e, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, p.ineq, p.ctx, bits=128)
            assert rc == -5 and st == ex.value.status, (seed, i, rc, st)
            continue
        assert r["status"] != pb.ST_ABORT, (seed, i)
        want = "void\n" if r["status"] == pb.ST_VOID else r["text"]
        assert pb.squash(text) == pb.squash(want) and piv == r["pivots"], (seed, i, piv, r["pivots"])
        checked += 1
    assert checked >= 12


REFPIP_GPU_GMP = os.path.join(os.path.dirname(pb.REFPIP_GPU), "refpip_gpu_gmp")


@pytest.mark.skipif(not os.access(REFPIP_GPU_GMP, os.X_OK), reason="oracle/_ref/refpip_gpu_gmp not built (no /root/reference or no gmp.h)")
@pytest.mark.parametrize("name", PIPTEST_DAT)
def test_dat_golden_through_the_gmp_flavour_front_end(name):
    """The reference's GMP flavour (tab_get, tape, sol_edit on mpz_t) with its traiter calls bound to the 128-bit engine
    by bindings/piplib_traiter_hook.c compiled with -DPIPLIB_INT_GMP (pipamd_traiter_hook_gmp over pipamd_traiter128):
    test/*.dat vs test/*.ll -- nothing overflows on these inputs, so the arbitrary-precision front end over the
    overflow-safe device flavour must print the int64 goldens."""
    import subprocess
    p = subprocess.run([REFPIP_GPU_GMP, "dat", os.path.join(G, "test", name + ".dat")], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    want = open(os.path.join(G, "test", name + ".ll"), encoding="latin-1").read()
    assert pb.squash(p.stdout.decode("latin-1")) == pb.squash(want)


def test_plain_c_stream_of_batches(tmp_path):
    """examples/batch_stream.c (built by __graft_entry__.build()): a stream of batches from plain C and one host thread
    through pipamd_batch_solve_async / pipamd_batch_poll -- every tableau finished, and the pivots it counts are those
    the Python binding counts for the same batches."""
    import subprocess
    import numpy as np
    import torch
    from piplib_amd import engine as eng, synth
    exe = os.path.join(pb.ROOT, "examples", "batch_stream")
    if not os.access(exe, os.X_OK):
        pytest.skip("examples/batch_stream not built")
    nb, B, nvar, ni, lanes, steps = 3, 1500, 127, 64, 4, 10
    batches = [synth.lexmin_batch(1000 + 7919 * b, B, nvar, ni) for b in range(nb)]
    path = str(tmp_path / "rows.bin")
    with open(path, "wb") as f:
        for r in batches:
            f.write(np.ascontiguousarray(r, dtype="<i8").tobytes())
    p = subprocess.run([exe, path, str(nb), str(B), str(nvar), str(ni), str(lanes), str(steps)], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    line = p.stdout.decode().strip().splitlines()[-1]
    e = eng.Engine(0)
    piv = []
    for r in batches:
        b = eng.Batch(e, r, nvar, 0, tflags=eng.T_INT)
        b.load()
        b.solve()
        piv.append(b.counters()["pivots"])
    want = sum(piv[k % nb] for k in range(steps))
    assert f": {want} pivots in" in line, (line, want)
    assert f"solution {lanes * B} nil 0 other 0" in line, line


@pytest.mark.parametrize("seed,shape", [(91, (5, 0, 7, 0)), (92, (5, 2, 7, 2)), (93, (8, 1, 10, 1)), (94, (4, 3, 9, 3)), (95, (12, 2, 20, 2))])
def test_dual_on_the_device_tree_vs_host_tree(seed, shape):
    """Compute_dual (TRAITER_DUAL, rational solves): the device-resident traiter() against the host tree (which the
    fixtures of the reference pin) on random problems with and without parameters -- same tape cells, same pivot count,
    and the device really served them."""
    from piplib_amd import engine as eng, synth
    probs = synth.random_problems(seed, 40, *shape, 0)
    e_dev, e_host = eng.Engine(0), eng.Engine(0)
    e_host.set_device_tree(False)
    served = compared = 0
    for p in probs:
        try:
            want = eng.traiter(e_host, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, eng.T_DUAL, p.ineq, p.ctx)
        except eng.SolverError as ex:
            with pytest.raises(eng.SolverError):
                eng.traiter(e_dev, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, eng.T_DUAL, p.ineq, p.ctx)
            continue
        got = eng.traiter(e_dev, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, eng.T_DUAL, p.ineq, p.ctx)
        served += e_dev.last_device_tree()[0]
        assert got == want
        compared += 1
    assert compared >= 30 and served >= 0.9 * compared, (compared, served)
