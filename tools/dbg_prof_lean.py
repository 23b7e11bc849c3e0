"""Diagnostic (not a test): where a wave of the lean bulk kernel (csrc/pip_lean.h) spends its cycles, per piece of the
pivot loop (`python -m piplib_amd.build --profile` build: s_memtime stamps, 4 waves per SIMD so that the stamps do not
spill).  python3 tools/dbg_prof_lean.py [tableaux [nvar,ni]] -- a small batch (e.g. 256) has every wave alone on its
SIMD: the latency chain itself; 10000 is the full GPU."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from piplib_amd import engine as eng, synth
eng.LIB_PATH = os.path.join(eng.HERE, "libpipamd_prof.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NVAR, NI = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "127,64").split(","))
INT = int(os.environ.get("INT", "1"))
rows = synth.lexmin_batch(1000 if NVAR == 127 else 2000, B, NVAR, NI)
e = eng.Engine(0)
e.set_bulk_min(1)
L = eng.lib()
L.pipamd_debug_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
b = eng.Batch(e, rows, NVAR, 0, tflags=(eng.T_INT if INT else 0) | eng.T_ROWS_STAY)
L.pipamd_debug_profile(e._h, 1, None)
e.debug_single_launch(2)   # the lean launch alone
e.set_timing(True)
for it in range(2):
    b.load(); b.solve()
    out = (C.c_uint64 * 64)()
    L.pipamd_debug_profile(e._h, 1, out)
    v = np.array(list(out), dtype=np.float64)[:16]
    c = b.counters()
    names = ["exam / integrer / cut", "pivot row load", "choisir_piv", "work list", "queue + recycled slot + barrier", "wait for a work row",
             "multipliers", "products + row gcd + division", "store + summary", "phase C", "entry", "epilogue"]
    print(f"lean launch {e.last_launch_ms(0):.3f} ms, pivots {c['pivots']} rows_rewritten {c['rows_rewritten']} cuts {c['cuts']} finished {c['finished']}")
    for n, x in zip(names, v):
        print(f"  {n:32s} {100 * x / v.sum():5.1f}%  {x / c['pivots']:9.0f} clocks/pivot")
    print(f"  {'all':32s}        {v.sum() / c['pivots']:9.0f} clocks/pivot")
