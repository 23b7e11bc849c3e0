"""What the lean kernel (csrc/pip_lean.h) does on the headline batch (GPU box): per-launch durations of one
un-pipelined solve, and tableaux finished / pivots done after the lean launch alone, after both bulk launches and
at the end.  python3 tools/lean_split.py [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from piplib_amd import engine as eng, synth
cfg = dict(bench.MAIN)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
rows = torch.as_tensor(synth.lexmin_batch(seed, cfg["batch"], cfg["nvar"], cfg["ni"], **cfg["gen"]), dtype=torch.int64).to(dev).contiguous()
e = eng.Engine(0)
e.set_max_rows(cfg["ni"] + 1024)
tf = (eng.T_INT if cfg["integer"] else 0) | eng.T_ROWS_STAY
b = eng.Batch(e, None, cfg["nvar"], 0, tflags=tf, entier_bits=cfg["ebits"], shape=tuple(rows.shape))
for lean in (0, 1):
    e.debug_lean(lean)
    e.set_timing(True)
    for rep in range(2):
        b.load_parts([rows]); b.solve()
    n = e.last_solve_launches()
    print("lean %d: launches %s ms, total %s" % (lean, [round(e.last_launch_ms(i), 3) for i in range(n)], b.counters()))
    for lvl in ((2, 1) if lean else (1,)):
        e.debug_single_launch(lvl)
        b.load_parts([rows]); b.solve()
        print("   stop level %d: %s" % (lvl, b.counters()))
        if lvl == 2:  # PipJob (csrc/pip_job.h, 200 bytes): status at 72, npiv at 80, the lean kernel's exit reason at 172
            j = b.ws[:25 * rows.shape[0]].view(torch.int32).view(rows.shape[0], 50).cpu()
            run = j[:, 18] == 0
            names = {0: "not taken", 1: "pivot budget", 2: "a row beyond class 0", 3: "cut denominator", 4: "pivot row denominator", 5: "LDS image full"}
            for w in range(6):
                m = run & (j[:, 43] == w)
                if int(m.sum()):
                    print("      still running, %-22s: %5d tableaux, pivots so far mean %.1f" % (names[w], int(m.sum()), float(j[m][:, 20].float().mean())))
        e.debug_single_launch(0)
