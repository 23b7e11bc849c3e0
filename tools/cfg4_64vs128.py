"""configs[4]'s tableaux through the 64-bit and the 128-bit engine, bulk launch only (96 pivots per tableau): how much
faster would 64-bit rows be while they fit?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from piplib_amd import engine as eng, synth
oc = [c for c in bench.OTHERS if c["key"] == "configs[4]"][0]
rows = synth.lexmin_batch(2000, 4000, oc["nvar"], oc["ni"], **oc["gen"])
for ebits in (64, 128):
    e = eng.Engine(0)
    e.set_max_rows(oc["ni"] + 1024)
    e.set_timing(True)
    b = eng.Batch(e, rows, oc["nvar"], 0, tflags=eng.T_INT | eng.T_ROWS_STAY if ebits == 64 else eng.T_INT, entier_bits=ebits)
    e.debug_single_launch(1)
    for rep in range(2):
        b.load(); b.solve(); torch.cuda.synchronize()
    c = b.counters()
    print("ebits", ebits, "bulk launch %.3f ms" % e.last_launch_ms(0), c, "-> %.1f M pivots/s in the launch" % (c["pivots"] / e.last_launch_ms(0) / 1e3))
