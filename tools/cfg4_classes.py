"""How large do the entries of configs[4]'s tableaux get (GPU box)?  The batch through the 128-bit engine; per tableau
the magnitude class of its largest entry when it ended (PipJob.maxabs: classes of the 128-bit kernel are 0: < 2^31,
1: < 2^63, 2: < 2^95, 3: beyond) and, stopped after the bulk launch (96 pivots), the same."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from piplib_amd import engine as eng, synth
oc = [c for c in bench.OTHERS if c["key"] == "configs[4]"][0]
rows = synth.lexmin_batch(2000, 4000, oc["nvar"], oc["ni"], **oc["gen"])
e = eng.Engine(0)
e.set_max_rows(oc["ni"] + 1024)
b = eng.Batch(e, rows, oc["nvar"], 0, tflags=eng.T_INT, entier_bits=128)
for lvl in (1, 0):
    e.debug_single_launch(lvl)
    b.load(); b.solve(); torch.cuda.synchronize()
    j = b.ws[:25 * rows.shape[0]].view(torch.int32).view(rows.shape[0], 50).cpu().numpy()
    status, npiv, mc = j[:, 18], j[:, 20], j[:, 40]   # maxabs (uint64) at byte 160: its low word
    print("stop level", lvl, "status", dict(zip(*np.unique(status, return_counts=True))), "class of the largest entry", dict(zip(*np.unique(mc, return_counts=True))), "pivots mean %.1f max %d" % (npiv.mean(), npiv.max()), b.counters())
