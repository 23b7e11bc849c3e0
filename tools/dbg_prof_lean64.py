"""Diagnostic (not a test): where a wave of the 128-bit flavour's lean kernel (csrc/pip_lean64.h) spends its cycles on
configs[4]'s pinned batch, per piece of the pivot loop (`python -m piplib_amd.build --profile` build).
python3 tools/dbg_prof_lean64.py [REP]  -- the batch REP times over (8: the GPU is full)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import make_bigint_fixtures as mk
from piplib_amd import engine as eng
eng.LIB_PATH = os.path.join(eng.HERE, "libpipamd_prof.so")
rows = np.concatenate([mk.rows_full("wide128")] * (int(sys.argv[1]) if len(sys.argv) > 1 else 1))
e = eng.Engine(0)
e.set_max_rows(128 + 1280)
L = eng.lib()
L.pipamd_debug_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
b = eng.Batch(e, rows, 255, 0, tflags=eng.T_INT, entier_bits=128)
L.pipamd_debug_profile(e._h, 1, None)
e.debug_single_launch(1)   # the lean launch alone
e.set_timing(True)
for it in range(2):
    b.load(); b.solve()
    out = (C.c_uint64 * 64)()
    L.pipamd_debug_profile(e._h, 1, out)
    v = np.array(list(out), dtype=np.float64)[:16]
    c = b.counters()
    names = ["exam / integrer / cut", "pivot row load", "choisir_piv", "work list", "recycled slot + barrier", "wait for a work row",
             "multipliers", "products + row gcd + division", "store + summary", "phase C", "entry", "epilogue"]
    print(f"lean64 launch {e.last_launch_ms(0):.3f} ms, pivots {c['pivots']} rows_rewritten {c['rows_rewritten']} cuts {c['cuts']} finished {c['finished']}")
    for n, x in zip(names, v):
        print(f"  {n:32s} {100 * x / v.sum():5.1f}%  {x / c['pivots']:9.0f} clocks/pivot")
    print(f"  {'all':32s}        {v.sum() / c['pivots']:9.0f} clocks/pivot")
