"""The parametric set of bench.py's `parametric` leg, N-fold, with the device tree's own timing lines
(GPU box):   PIPAMD_FOREST_STATS=1 python tools/dbg_param.py [N=8]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from piplib_amd import engine as eng, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = json.load(open(os.path.join(ROOT, "tests", "manual", "forest_good.json")))
allp = synth.random_problems(cfg["seed"], cfg["count"], *cfg["shape"], 1, cmax=cfg["cmax"], bmax=cfg["bmax"])
probs = [allp[i] for i in cfg["good"]] * n
e = eng.Engine(0)
for rep in range(3):
    prep = eng.PreparedProblems(probs)
    t0 = time.perf_counter(); eng.solve_prepared(e, prep, lockstep=True); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); r = prep.results(); dr = time.perf_counter() - t1
    print(f"{len(probs)} problems: C call {dt*1e3:.2f} ms = {len(probs)/dt:.0f} problems/s; text {dr*1e3:.1f} ms; device tree {e.last_device_tree()}", flush=True)
