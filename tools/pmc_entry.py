"""One un-pipelined solve of the headline batch with ITER pivots per launch and WAVES waves per
tableau (for tools/pmc_entry.sh).  FLAGS: 1 = integer solve, 0 = rational."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from piplib_amd import engine as eng, synth
rows = torch.as_tensor(synth.lexmin_batch(1000, 10000, 127, 64)).to("cuda:0")
e = eng.Engine(0)
it = int(os.environ.get("ITER", "1000000"))
e.set_iter_limit(it)
if os.environ.get("WAVES"):
    e.set_waves_per_job(int(os.environ["WAVES"]))
e.set_round_pivots(it if it < 1000000 else 1000000)
e.set_round_rows(128)
b = eng.Batch(e, rows, 127, 0, tflags=int(os.environ.get("FLAGS", "1")))
b.load(); b.solve()
torch.cuda.synchronize()
c = b.counters(); b.fetch()
p = b.pivots.cpu().numpy()
print("RUN iter", it, "launches", e.last_solve_launches(), "pivots", c["pivots"], "rows", c["rows_rewritten"], "cuts", c["cuts"],
      "entries", int(np.ceil(p / it).sum()) if it < 1000000 else len(p))
