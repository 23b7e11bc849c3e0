"""What the lean kernel of the 128-bit flavour (csrc/pip_lean64.h) does on configs[4]'s pinned batch (GPU box): the lean
launch alone (pipamd_debug_single_launch) -- its duration, tableaux finished, pivots done, and why tableaux left it
(PipJob.pad_: 1 pivot budget, 2 a row beyond long longs, 3 a cut's denominator, 5 no room in the LDS image)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import make_bigint_fixtures as mk
from piplib_amd import engine as eng
rows = mk.rows_full("wide128")
e = eng.Engine(0)
e.set_max_rows(128 + 1280)
e.set_timing(True)
b = eng.Batch(e, rows, 255, 0, tflags=eng.T_INT, entier_bits=128)
e.debug_single_launch(1)
for _ in range(2):
    b.load(); b.solve(); torch.cuda.synchronize()
B = rows.shape[0]
j = b.ws[:25 * B].view(torch.int32).view(B, 50).cpu().numpy()
status, npiv, why, mcls, ni = j[:, 18], j[:, 20], j[:, 43], j[:, 40], j[:, 10]
c = b.counters()
print("lean64 launch %.2f ms: %d pivots of the batch's 311783, finished %d of %d; statuses %s" % (e.last_launch_ms(0), c["pivots"], c["finished"], B, dict(zip(*np.unique(status, return_counts=True)))))
run = status == eng.ST_RUN
print("left running: why %s; pivots done there: mean %.1f; rows mean %.1f; classes %s" % (dict(zip(*np.unique(why[run], return_counts=True))), npiv[run].mean() if run.any() else 0, ni[run].mean() if run.any() else 0, dict(zip(*np.unique(mcls[run], return_counts=True)))))
for w in np.unique(why[run]):
    m = run & (why == w)
    print("  why %d: %d tableaux, pivots done min/median/max %d/%d/%d, rows %d..%d" % (w, m.sum(), npiv[m].min(), np.median(npiv[m]), npiv[m].max(), ni[m].min(), ni[m].max()))
