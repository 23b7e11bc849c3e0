"""Manual GPU debug driver (not a test): python tests/dbg_gpu.py nvar ni batch nq"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pipbatch as pb
from gpu_common import *

nvar, ni, B, nq = [int(x) for x in sys.argv[1:5]]
rows = synth.lexmin_batch(5, B, nvar, ni)
o = oracle_batch(rows, nvar, 0, nq)
g = gpu_batch(rows, nvar, 0, nq, iter_limit=5000)
st = g.status.cpu().numpy(); pv = g.pivots.cpu().numpy(); cu = g.cuts.cpu().numpy()
num = g.sol_num.cpu().numpy(); den = g.sol_den.cpu().numpy()
bad = 0
for b, r in enumerate(o.results):
    want = pb.squash(r.text)
    got = pb.squash(solution_text(num[b], den[b])) if st[b] == 1 else ("()" if st[b] == 2 else f"status{st[b]}")
    ok = got == want and pv[b] == r.pivots
    if not ok:
        bad += 1
        if bad < 6:
            print(f"MISMATCH b={b} st={st[b]} piv gpu={pv[b]} cpu={r.pivots} cuts={cu[b]}\n  got  {got[:160]}\n  want {want[:160]}")
print(f"shape nvar={nvar} ni={ni} B={B} nq={nq}: mismatches {bad}/{B}; pivots gpu {pv.sum()} cpu {o.total_pivots}; kernel {g.last_solve_ms():.3f} ms")
