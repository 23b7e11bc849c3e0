"""Diagnostic (not a test): per-phase cycle shares of pip_advance_kernel, -DPIP_PROFILE build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from piplib_amd import engine as eng, synth
EVENTS = int(os.environ.get("EVENTS", "0"))  # 1: event counters (python -m piplib_amd.build --profile-events)
eng.LIB_PATH = os.path.join(eng.HERE, "libpipamd_prof_events.so" if EVENTS else "libpipamd_prof.so")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
EBITS = int(os.environ.get("EBITS", "64"))
NVAR, NI = (int(x) for x in os.environ.get("SHAPE", "127,64").split(","))
rows = synth.lexmin_batch(1000 if NVAR == 127 else 77, B, NVAR, NI)
if os.environ.get("GEN") == "wide128":  # configs[4]'s pinned family (EBITS=128 SHAPE=255,128); HARD=n: its n longest tableaux only
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import json, make_bigint_fixtures as mk
    rows = mk.rows_full("wide128")
    if os.environ.get("HARD"):
        rec = json.load(open(os.path.join(os.path.dirname(mk.__file__), "gmp", "wide128.json")))["problems"]
        rows = rows[np.sort(np.argsort([r["pivots"] for r in rec])[-int(os.environ["HARD"]):])]
    rows = rows[:B]
e = eng.Engine(0)
if os.environ.get("MAXROWS"): e.set_max_rows(int(os.environ["MAXROWS"]))
if len(sys.argv) > 2 and int(sys.argv[2]): e.set_waves_per_job(int(sys.argv[2]))
if len(sys.argv) > 3: e.set_round_pivots(int(sys.argv[3]))
L = eng.lib()
L.pipamd_debug_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
b = eng.Batch(e, rows, NVAR, 0, tflags=eng.T_INT, entier_bits=EBITS)
L.pipamd_debug_profile(e._h, 1, None)
for it in range(2):
    b.load(); b.solve()
    out = (C.c_uint64 * 64)()
    L.pipamd_debug_profile(e._h, 1, out)
    ms = b.last_solve_ms()
    v = np.array(list(out), dtype=np.float64)
    c = b.counters()
    names = ["entry", "exam", "integrer", "A prow+guard", "A column", "A worklist+det", "B rest", "C flags", "epilogue", "B load wait", "B multipliers", "B update_row", "B store+publish", "entry tables", "entry pass", "entry sort"]
    print(f"launches {e.last_solve_launches()} kernel {ms:.2f} ms pivots {c['pivots']} rows_rewritten {c['rows_rewritten']} cuts {c['cuts']}")
    cn = ["rows updated", "pivot != 1", "gcd(pivot, foo) != 1", "g0 != 1", "row divided (g != 1)", "den != 1", "pivot >= 2^16", "gcd_u64 calls", "gcd_u64 iterations", "gcd_u32 calls", "gcd_u32 iterations", "cquo 32-bit divisions", "cquo 64-bit divisions", "refinement rounds", "refinement rounds (64-bit mod)", "choose: ratio-loop passes", "choose: rows loaded", "choose: rows considered", "choose: 64-row blocks", "choose: single candidate", "128-bit binary gcds", "128-bit binary gcd iterations"]
    for n, x in zip(cn, v[16:63] if EVENTS else []):
        print(f"  {n:24s} {x / c['pivots']:7.3f} per pivot")
    v = v[:16]
    for n, x in zip(names, v):
        print(f"  {n:10s} {100*x/v.sum():5.1f}%  {x/c['pivots']:9.0f} cycles/pivot")
