"""One un-pipelined integer solve of the headline batch (for rocprofv3 --pmc runs, tools/pmc_dup.sh)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piplib_amd import engine as eng, synth
rows = torch.as_tensor(synth.lexmin_batch(1000, 10000, 127, 64)).to("cuda:0")
e = eng.Engine(0)
if os.environ.get("ROUND"):
    e.set_round_pivots(int(os.environ["ROUND"]))
b = eng.Batch(e, rows, 127, 0, tflags=eng.T_INT)
b.load(); b.solve()
torch.cuda.synchronize()
c = b.counters()
b.fetch()
print("RUN", os.path.basename(eng.LIB_PATH), "launches", e.last_solve_launches(), c["pivots"], c["rows_rewritten"], c["cuts"], int(b.status.sum().item()), int(b.pivots.sum().item()))
