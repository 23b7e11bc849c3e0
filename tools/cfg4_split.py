"""Where does a lone configs[4] batch (the pinned `wide128` family) spend its time (GPU box)?  Per-launch durations of one
un-pipelined solve of the whole batch, then of the batch without its N longest tableaux (by the reference-GMP pivot
counts of tests/golden/gmp/wide128.json) and of those N alone: how much of the step is the latency chain of a few
slow-converging tableaux.  python3 tools/cfg4_split.py [N [tail waves]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import make_bigint_fixtures as mk
from piplib_amd import engine as eng

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
TW = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = mk.rows_full("wide128")
REP = int(os.environ.get("REP", "1"))  # the batch REP times over (a larger batch of the same tableaux)
rec = json.load(open(os.path.join(ROOT, "tests", "golden", "gmp", "wide128.json")))["problems"]
piv = np.array([r["pivots"] for r in rec])
order = np.argsort(piv)
hard, easy = np.sort(order[-N:]), np.sort(order[:-N])


def run(name, r):
    e = eng.Engine(0)
    e.set_max_rows(128 + 1280)
    if TW:
        e.set_tail_waves(TW)
    # engine knobs from the environment: WAVES (waves per tableau, every launch), BULKMIN, ROUND (pivot budget of the
    # one-wave bulk launches), ROUNDROWS (spare rows of their LDS image)
    if os.environ.get("WAVES"): e.set_waves_per_job(int(os.environ["WAVES"]))
    if os.environ.get("BULKMIN"): e.set_bulk_min(int(os.environ["BULKMIN"]))
    if os.environ.get("ROUND"): e.set_round_pivots(int(os.environ["ROUND"]))
    if os.environ.get("ROUNDROWS"): e.set_round_rows(int(os.environ["ROUNDROWS"]))
    e.set_timing(True)
    b = eng.Batch(e, r, 255, 0, tflags=eng.T_INT, entier_bits=128)
    for it in range(2):
        b.load(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.solve(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    c = b.counters()
    b.fetch(); torch.cuda.synchronize()
    pv, ct = b.pivots.cpu().numpy(), b.cuts.cpu().numpy()
    nl = e.last_solve_launches()
    print("%-28s %5d tableaux: %9.2f ms, %8d pivots (max %5d), cuts max %4d, rows rewritten per pivot %.1f -> %.2f M pivots/s; launches: %s"
          % (name, r.shape[0], dt * 1e3, c["pivots"], pv.max(), ct.max(), c["rows_rewritten"] / max(1, c["pivots"]), c["pivots"] / dt / 1e6,
             " ".join("%.1f" % e.last_launch_ms(i) for i in range(nl))), flush=True)
    del b
    e.close()


print("pivots of the %d longest: %s" % (N, sorted(piv[hard].tolist())))
run("all", np.concatenate([rows] * REP))
run("without the %d longest" % N, np.concatenate([rows[easy]] * REP))
run("the %d longest alone" % N, rows[hard])
run("the longest alone", rows[order[-1:]])
