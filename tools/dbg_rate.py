"""Manual GPU measurement: pivots/s for variations of the headline workload."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from piplib_amd import engine as eng, synth
dev = torch.device("cuda", 0)
def rate(rows, nvar, nq, depth=3, steps=9, **kw):
    rows_d = torch.as_tensor(rows).to(dev)
    lanes = []
    for _ in range(depth):
        e = eng.Engine(0)
        if os.environ.get("WAVES"):
            e.set_waves_per_job(int(os.environ["WAVES"]))
        lanes.append((e, eng.Batch(e, rows_d, nvar, 0, tflags=eng.T_INT if nq else 0, **kw), torch.cuda.Stream(dev)))
    def work(i, n):
        with torch.cuda.stream(lanes[i][2]):
            for _ in range(n):
                lanes[i][1].load(); lanes[i][1].solve()
            lanes[i][2].synchronize()
    for i in range(depth): work(i, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i, steps // depth)) for i in range(depth)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = lanes[0][1].counters()
    return c["pivots"] * steps / dt / 1e6, c, dt / steps * 1e3
rows = synth.lexmin_batch(1000, 10000, 127, 64)
for nq in (1, 0):
    r, c, ms = rate(rows, 127, nq)
    print(f"64x128 nq={nq}: {r:.1f} Mpiv/s  {ms:.2f} ms/step  pivots {c['pivots']} cuts {c['cuts']} rows/pivot {c['rows_rewritten']/c['pivots']:.2f}", flush=True)
rows = synth.lexmin_batch(1000, 40000, 63, 32)
r, c, ms = rate(rows, 63, 1)
print(f"32x64 x40k nq=1: {r:.1f} Mpiv/s  {ms:.2f} ms/step  pivots {c['pivots']}", flush=True)
rows = synth.lexmin_batch(1000, 1000, 63, 32)
r, c, ms = rate(rows, 63, 0)
print(f"32x64 x1k nq=0 (BASELINE configs[1]): {r:.1f} Mpiv/s  {ms:.3f} ms/step  pivots {c['pivots']}", flush=True)
rows = synth.lexmin_batch(77, 1000, 255, 128)
r, c, ms = rate(rows, 255, 1, entier_bits=128)
print(f"128x256 x1k nq=1 int128 (BASELINE configs[4]): {r:.1f} Mpiv/s  {ms:.3f} ms/step  pivots {c['pivots']} cuts {c['cuts']}", flush=True)
r, c, ms = rate(rows, 255, 1, entier_bits=64)
print(f"128x256 x1k nq=1 int64: {r:.1f} Mpiv/s  {ms:.3f} ms/step  pivots {c['pivots']}", flush=True)
