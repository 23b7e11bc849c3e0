"""Per-launch durations of pipamd_batch_solve on the bench workload (run on the GPU box).
   python tools/dbg_launches.py [round_pivots] [round_rows] [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piplib_amd import engine as eng, synth
K1 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
KA = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
rows = torch.as_tensor(synth.lexmin_batch(1000, B, 127, 64)).cuda()
e = eng.Engine(0)
if K1: e.set_round_pivots(K1)
if KA: e.set_round_rows(KA)
b = eng.Batch(e, rows, 127, 0, tflags=eng.T_INT)
L = eng.lib()
L.pipamd_last_launch_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
for it in range(3):
    b.load(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); b.solve(); t1.record(); torch.cuda.synchronize()
    ms = []
    for i in range(e.last_solve_launches()):
        v = C.c_float(); L.pipamd_last_launch_ms(e._h, i, C.byref(v)); ms.append(round(v.value, 3))
    print("K1", K1, "KA", KA, "solve %.3f ms" % t0.elapsed_time(t1), "launches", ms, flush=True)
print(b.counters())
