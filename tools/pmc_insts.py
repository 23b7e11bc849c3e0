"""Manual GPU measurement under `rocprofv3 --pmc SQ_INSTS_*`: one un-pipelined solve each of
(a) the integer headline batch, (b) the same batch solved rationally (no cuts), (c) the integer
batch with row skipping off.  tools/pmc_insts_report.py turns the counter CSV into instructions
per pivot and per rewritten row."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piplib_amd import engine as eng, synth
rows = torch.as_tensor(synth.lexmin_batch(1000, 10000, 127, 64)).to("cuda:0")
out = []
for name, flags, waves in (("integer", eng.T_INT, 0), ("rational", 0, 0), ("integer_noskip", eng.T_INT | eng.T_NOSKIP, 4)):
    e = eng.Engine(0)
    if waves:
        e.set_waves_per_job(waves)
    b = eng.Batch(e, rows, 127, 0, tflags=flags)
    b.load(); b.solve()
    torch.cuda.synchronize()
    c = b.counters()
    out.append({"name": name, "pivots": c["pivots"], "rows_rewritten": c["rows_rewritten"], "cuts": c["cuts"],
                "launches": e.last_solve_launches()})
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "pmc_insts_runs.json"), "w"))
print(out)
