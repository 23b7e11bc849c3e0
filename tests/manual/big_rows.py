"""Manual GPU check: tableaux with many rows (up to the 2048-slot / LDS-budget limit) vs the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from piplib_amd import synth
import gpu_common as gc
for nvar, ni, nq, cap in ((20, 900, 1, 300), (100, 1500, 1, 400), (300, 1000, 1, 200), (510, 600, 0, 8), (60, 1900, 0, 100), (20, 3200, 0, 100)):
    rows = synth.lexmin_batch(5, 4, nvar, ni)
    n, piv = gc.compare(rows, nvar, 0, nq, cap_cuts=cap)
    print(f"nvar={nvar} ni={ni} nq={nq}: {n} tableaux, {piv} pivots bit-exact", flush=True)
