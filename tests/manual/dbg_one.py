"""Manual GPU check: one problem of the dbg_forest set through the device tree vs the forest vs the CPU."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pipbatch as pb
from piplib_amd import engine as eng, synth
cfg = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "forest_good.json")))
allp = synth.random_problems(cfg["seed"], cfg["count"], *cfg["shape"], 1, cmax=cfg["cmax"], bmax=cfg["bmax"])
probs = [allp[i] for i in cfg["good"]]
which = [int(x) for x in sys.argv[1:]] or [220]
sel = [probs[i] for i in which]
exe = pb.REFPIP if pb.have_ref() else pb.ORACLEPIP
o = pb.run_batch(exe, sel, pb.F_NOTEXT)
print("cpu pivots", o.total_pivots, [r.pivots for r in o.results] if hasattr(o, "results") else "")
e = eng.Engine(0)
d = eng.solve_tableaux(e, sel, lockstep=True)
print("device tree", e.last_device_tree(), [(x[1], x[2], x[3]) for x in d])
