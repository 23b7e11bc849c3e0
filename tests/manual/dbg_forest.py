"""Manual GPU measurement: lock-step Forest vs threaded trees on mid-size parametric problems."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pipbatch as pb
from piplib_amd import engine as eng, synth
cfg = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "forest_good.json")))
allp = synth.random_problems(cfg["seed"], cfg["count"], *cfg["shape"], 1, cmax=cfg["cmax"], bmax=cfg["bmax"])
probs = [allp[i] for i in cfg["good"]]
exe = pb.REFPIP if pb.have_ref() else pb.ORACLEPIP
o = pb.run_batch(exe, probs, pb.F_NOTEXT)
print(f"{len(probs)} problems shape {cfg['shape']}; CPU 1 core ({os.path.basename(exe)}): {o.solve_seconds*1e3:.1f} ms = {len(probs)/o.solve_seconds:.0f} problems/s, {o.total_pivots} pivots", flush=True)
e = eng.Engine(0)
eng.solve_tableaux(e, probs[:30], lockstep=True)
for rep in range(3):
    prep = eng.PreparedProblems(probs)
    t = time.perf_counter(); eng.solve_prepared(e, prep, lockstep=True); td = time.perf_counter() - t
    d = prep.results()
    print(f"device tree: {td*1e3:.1f} ms (the C call)  {len(probs)/td:.0f} problems/s = {len(probs)/td/(len(probs)/o.solve_seconds):.1f} x one CPU core  pivots {sum(x[3] for x in d)}  (served, handed back) = {e.last_device_tree()}", flush=True)
e.set_device_tree(False)
prep = eng.PreparedProblems(probs)
t = time.perf_counter(); eng.solve_prepared(e, prep, lockstep=True); tb = time.perf_counter() - t
b = prep.results()
print(f"lockstep: {tb*1e3:.1f} ms  {len(probs)/tb:.0f} problems/s  pivots {sum(x[3] for x in b)}", flush=True)
print("device tree vs lockstep mismatches", sum(x != y for x, y in zip(d, b)), [i for i, (x, y) in enumerate(zip(d, b)) if x != y][:10])
t = time.perf_counter(); a = eng.solve_tableaux(e, probs, nthreads=16); ta = time.perf_counter() - t
print(f"threads : {ta*1e3:.1f} ms  {len(probs)/ta:.0f} problems/s", flush=True)
print("mismatches", sum(x != y for x, y in zip(a, b)))
