"""Manual GPU debug: random parametric problems one by one with progress output."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pipbatch as pb
from piplib_amd import engine as eng, synth
seed, nvar, nparm, ni, nc, nq, deepest = [int(x) for x in sys.argv[1:8]]
lo = int(sys.argv[8]) if len(sys.argv) > 8 else 0
hi = int(sys.argv[9]) if len(sys.argv) > 9 else 40
probs = synth.random_problems(seed, 40, nvar, nparm, ni, nc, nq)
o = pb.run_batch(pb.ORACLEPIP, probs, pb.F_DEEPEST if deepest else 0)
e = eng.Engine(0)
for i in range(lo, hi):
    p, r = probs[i], o.results[i]
    t = time.time()
    print(f"case {i}: oracle status {r.status} pivots {r.pivots} ...", flush=True)
    try:
        text, piv = eng.solve_tableau(e, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, p.ineq, p.ctx, True, bool(deepest))
        ok = (pb.squash(text) == ("void" if r.status == pb.ST_VOID else pb.squash(r.text))) and piv == r.pivots
        print(f"   gpu pivots {piv} ok={ok} {time.time()-t:.3f}s", flush=True)
        if not ok:
            print("   GOT ", pb.squash(text)[:300]); print("   WANT", pb.squash(r.text)[:300], flush=True)
    except eng.SolverError as ex:
        print(f"   gpu SolverError status {ex.status} (oracle abort={r.status == pb.ST_ABORT})", flush=True)
