"""-m gpu: PipLib's own test inputs through the host decision tree + HIP engine
(pipamd_solve_tableau, layer 3 of the C ABI) against the reference's golden .ll files."""
import json
import os

import pytest

import pipbatch as pb
from datfile import read_dat
from test_oracle_golden import PIPTEST_DAT, MANIFEST

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
G = os.path.join(pb.ROOT, "tests", "golden")


def solve_file(path, **kw):
    from piplib_amd import engine as eng
    e = eng.Engine(0)
    out = []
    status = None
    for p in read_dat(path):
        out.append("(" + p["comment"])
        try:
            text, _ = eng.solve_tableau(e, p["nvar"], p["nparm"], p["ni"], p["nc"], p["bigparm"], p["nq"],
                                        p["ineq"], p["ctx"], **kw)
        except eng.SolverError as ex:
            status = ex.status
            break
        out.append(text if text == "void\n" else ")\n" + text)
        out.append(")\n")
    return "".join(out), status


@pytest.mark.parametrize("name", PIPTEST_DAT)
def test_dat_golden_on_gpu(name):
    got, status = solve_file(os.path.join(G, "test", name + ".dat"))
    assert status is None
    want = open(os.path.join(G, "test", name + ".ll"), encoding="latin-1").read()
    assert pb.squash(got) == pb.squash(want)


@pytest.mark.parametrize("key", sorted(MANIFEST))
def test_ref_generated_on_gpu(key):
    m = MANIFEST[key]
    got, status = solve_file(os.path.join(G, key))
    want = open(os.path.join(G, "ref_dp", m["ll"]), encoding="latin-1").read()
    if m["rc"] != 0:  # the reference exit(1)s with "Integer overflow": same verdict, same partial output
        from piplib_amd import engine as eng
        assert status == eng.ST_OVERFLOW
    else:
        assert status is None
    assert pb.squash(got) == pb.squash(want)


from test_oracle_golden import PIPTEST_PIP  # noqa: E402


@pytest.mark.parametrize("name", PIPTEST_PIP)
def test_pip_solve_golden_on_gpu(name):
    """example/*.pip through pipamd_pip_solve (PolyLib matrices in, PipQuast out) vs example/*.ll."""
    from datfile import read_pip, matrix_text
    from piplib_amd import engine as eng
    context, bignum, domain, opts = read_pip(os.path.join(G, "example", name + ".pip"))
    e = eng.Engine(0)
    bg = bignum + (domain.shape[1] - context.shape[1]) if bignum > 0 else bignum
    text, _ = eng.pip_solve(e, domain, context, bg, **opts)
    got = ("[PIP2-like future input] Please enter:\n- the context matrix,\n" + matrix_text(context) +
           "- the bignum column (start at 0, -1 if no bignum),\n" + f"{bignum}\n" +
           "- the constraint matrix.\n" + matrix_text(domain) + "\n" + text)
    want = open(os.path.join(G, "example", name + ".ll"), encoding="latin-1").read()
    assert pb.squash(got) == pb.squash(want)


@pytest.mark.parametrize("name", ["small", "square", "max", "big", "cg1", "sven"])
def test_compute_dual_on_gpu(name):
    """pip_solve with Nq = 0 and Compute_dual = 1 vs reference-generated fixtures."""
    from datfile import read_pip, matrix_text
    from piplib_amd import engine as eng
    d = os.path.join(G, "ref_dp")
    context, bignum, domain, opts = read_pip(os.path.join(d, f"dual__{name}.pip"))
    assert opts.get("Nq") == 0 and opts.get("Compute_dual") == 1
    e = eng.Engine(0)
    bg = bignum + (domain.shape[1] - context.shape[1]) if bignum > 0 else bignum
    text, _ = eng.pip_solve(e, domain, context, bg, **opts)
    got = ("[PIP2-like future input] Please enter:\n- the context matrix,\n" + matrix_text(context) +
           "- the bignum column (start at 0, -1 if no bignum),\n" + f"{bignum}\n" +
           "- the constraint matrix.\n" + matrix_text(domain) + "\n" + text)
    want = open(os.path.join(d, f"dual__{name}.ll"), encoding="latin-1").read()
    assert pb.squash(got) == pb.squash(want)


def test_plain_c_example():
    """examples/solve_small.c (built by __graft_entry__.build()): the C ABI from plain C --
    pip_solve drop-in output equals the reference's example/small.ll quast, the tableau-form
    call equals the oracle."""
    import subprocess
    import numpy as np
    from piplib_amd import synth
    exe = os.path.join(pb.ROOT, "examples", "solve_small")
    if not os.access(exe, os.X_OK):
        import __graft_entry__
        __graft_entry__.build()
    p = subprocess.run([exe], capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    out = pb.squash(p.stdout.decode())
    want_quast = pb.squash(open(os.path.join(G, "example", "small.ll")).read()).split("3")[-1]  # "(list#[0]#[0])"
    assert out.startswith(want_quast) or want_quast in out
    prob = synth.Problem(2, 1, 3, 1, -1, 1, np.array([[1, 1, 0, -1], [-1, 0, 5, 0], [0, -1, 7, 0]], dtype=np.int64),
                         np.array([[-1, 12]], dtype=np.int64))
    o = pb.run_batch(pb.ORACLEPIP, [prob])
    assert out.endswith(pb.squash(o.results[0].text))


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
def test_pip_solve_fuzz_regressions(name, dom, ctx, opts):
    """pipamd_pip_solve vs the oracle's `pip` mode on inputs that once failed with PIPAMD_ST_CAPACITY."""
    import subprocess
    import numpy as np
    from datfile import matrix_text
    from piplib_amd import engine as eng
    dom, ctx = np.array(dom, dtype=np.int64), np.array(ctx, dtype=np.int64)
    words = "".join(k + "\n" for k in opts)
    txt = (matrix_text(ctx) + "\n-1\n\n" + matrix_text(dom) + "\n" + words).encode()
    o = subprocess.run([pb.ORACLEPIP, "pip"], input=txt, capture_output=True, timeout=120)
    assert o.returncode == 0
    text, _ = eng.pip_solve(eng.Engine(0), dom, ctx, -1, **opts)
    got = ("[PIP2-like future input] Please enter:\n- the context matrix,\n" + matrix_text(ctx) +
           "- the bignum column (start at 0, -1 if no bignum),\n-1\n- the constraint matrix.\n" +
           matrix_text(dom) + "\n" + text)
    assert pb.squash(got) == pb.squash(o.stdout.decode("latin-1"))


def test_pip_solve_refuses_dual_with_urs_parms():
    """Compute_dual + Urs_parms makes the reference's sol_vector_edit take a negative-length vector
    (it exits with "Memory Overflow" or faults); the library must refuse the call, not crash."""
    import numpy as np
    from piplib_amd import engine as eng
    dom = np.array([[1, 3, -3, -3, 2, -2, -2, 4], [1, -3, 1, -3, 0, 1, -3, 10], [1, 1, 0, 2, -1, 2, -1, -5],
                    [1, 2, 3, 2, -2, -1, 3, 10], [1, 0, 3, 1, -1, -2, 0, -5], [1, -3, 0, 2, -3, -2, 0, -3]], dtype=np.int64)
    for npar in (1, 2, 3):
        ctx = np.zeros((0, npar + 2), dtype=np.int64)
        with pytest.raises(RuntimeError):
            eng.pip_solve(eng.Engine(0), dom, ctx, -1, Urs_unknowns=1, Urs_parms=1, Nq=0, Compute_dual=1)
