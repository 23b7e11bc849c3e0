"""The lean launch alone with a pivot budget of ROUND per tableau (for rocprofv3 --pmc: tools/pmc_lean_fit.sh fits
instructions = a * tableaux + b * pivots from several budgets)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piplib_amd import engine as eng, synth
rows = torch.as_tensor(synth.lexmin_batch(1000, 10000, 127, 64)).to("cuda:0")
e = eng.Engine(0)
e.set_round_pivots(int(os.environ.get("ROUND", "96")))
e.debug_single_launch(2)
b = eng.Batch(e, rows, 127, 0, tflags=eng.T_INT | eng.T_ROWS_STAY)
b.load(); b.solve()
torch.cuda.synchronize()
c = b.counters()
print("RUN round", os.environ.get("ROUND", "96"), "pivots", c["pivots"], "rows", c["rows_rewritten"], "cuts", c["cuts"], "finished", c["finished"])
