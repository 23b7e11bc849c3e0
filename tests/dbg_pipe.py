"""Experiment: two batches in flight on two streams/threads (step pipelining)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from piplib_amd import engine as eng, synth
rows = synth.lexmin_batch(1000, 10000, 127, 64)
dev = torch.device("cuda", 0)
rows_d = torch.as_tensor(rows).to(dev)
def mk():
    e = eng.Engine(0)
    return e, eng.Batch(e, rows_d, 127, 0, tflags=eng.T_INT)
for depth in (1, 2, 3):
    objs = [mk() for _ in range(depth)]
    streams = [torch.cuda.Stream(dev) for _ in range(depth)]
    def work(i, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                objs[i][1].load(); objs[i][1].solve()
            streams[i].synchronize()
    for i in range(depth): work(i, 1)
    torch.cuda.synchronize()
    K = 12
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i, K // depth)) for i in range(depth)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = objs[0][1].counters()
    print(f"depth {depth}: {dt/K*1e3:.3f} ms/step  {c['pivots']*K/dt/1e6:.1f} Mpiv/s finished {c['finished']}", flush=True)
