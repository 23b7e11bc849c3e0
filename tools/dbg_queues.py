"""Which torch streams share a HIP hardware queue? (GPU box)  A long spin kernel on stream a, a tiny
kernel on stream b: if b's kernel waits for a's, the two streams sit on one hardware queue.
   python tools/dbg_queues.py [N=24]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
x = torch.zeros(1, device=dev)
st = [torch.cuda.Stream(dev) for _ in range(n)]
for s in st:
    with torch.cuda.stream(s):
        x + 1
torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize(); long_ms = (time.perf_counter() - t0) * 1e3
print(f"spin kernel {long_ms:.2f} ms; GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}")
group = list(range(n))
for a in range(n):
    for b in range(a + 1, n):
        t0 = time.perf_counter()
        with torch.cuda.stream(st[a]):
            torch.cuda._sleep(20_000_000)
        with torch.cuda.stream(st[b]):
            x + 1
        st[b].synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        if dt > 0.5 * long_ms:
            group[b] = group[a]
print("queue classes:", group)
print("distinct:", len(set(group)))
