"""Where does a lone 1k batch of 32x64 tableaux spend its step (GPU box)?  Per-step wall time of load + solve + results in a
plain loop, with the synchronous and the asynchronous solve, with and without the events around the launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piplib_amd import engine as eng, synth
rows = torch.as_tensor(synth.lexmin_batch(2000, 1000, 63, 32)).to("cuda:0")
for timing, bulk in ((True, 0), (False, 0), (False, 256), (False, 257)):
    for mode in ("sync", "async"):
        e = eng.Engine(0)
        e.set_timing(timing)
        if bulk:
            e.set_bulk_min(256)
            e.set_lone_batches(bulk == 256)
        b = eng.Batch(e, None, 63, 0, tflags=eng.T_ROWS_STAY, shape=tuple(rows.shape))
        st = torch.cuda.Stream()
        def step():
            b.load_parts([rows], st.cuda_stream)
            if mode == "sync":
                b.solve(st.cuda_stream)
            else:
                b.solve_async(st.cuda_stream)
                while b.poll() == 0:
                    pass
            b.fetch(st.cuda_stream)
        for _ in range(10): step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 300
        for _ in range(n): step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("events %s bulk %s  %5s solve: %.3f ms per step, %.1f M pivots/s" % (timing, bulk, mode, dt / n * 1e3, b.counters()["pivots"] * n / dt / 1e6))
