"""Does a configuration's pipelined figure depend on what ran before it in the process? (GPU box)
   python tools/dbg_seq.py 1:24 4:12 4:12    runs the CFG:DEPTH items in order, one process"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
a = types.SimpleNamespace(waves=0, bulk_min=-1, round=0, round_rows=0)
dev = torch.device("cuda", 0)
def barrier(): torch.cuda.synchronize(dev)
last = None
for item in sys.argv[1:]:
    if item == 'G':
        from piplib_amd import dist as pdist
        print(pdist.gather_totals([1, 2, 3], 0.1, dev), last[1].counters()['pivots'], flush=True)
        continue
    if item == 'K':  # a timed solve of the previous item's lane 0 on the null stream (bench.kernel_ms_of)
        print('null-stream solve', bench.kernel_ms_of(last[1]), 'ms', flush=True)
        continue
    f = item.split(":")
    cfgi, depth = int(f[0]), int(f[1])
    stag = float(f[2]) if len(f) > 2 else -1.0  # CFG:DEPTH[:STAGGER[:STEPS]]
    cfg = {1: bench.OTHERS[0], 4: bench.OTHERS[1], 2: bench.MAIN}[cfgi]
    lanes = bench.Lanes(cfg, depth, dev, 0, [2000 + 7919 * i for i in range(depth)], a)
    if os.environ.get('DBG_FRESH'):  # fresh torch pool streams per item (the order effect bench.lane_stream removes)
        lanes.lanes = [(e, b, torch.cuda.Stream(dev)) for e, b, _ in lanes.lanes]
    steps = int(f[3]) if len(f) > 3 else 16 * depth
    dt, share = bench.timed(lanes, steps, depth, barrier, stag)
    tot = lanes.totals(share)
    print(f"{cfg['key']} depth {depth}: {tot[0] / dt / 1e6:.1f} Mpiv/s  {dt / steps * 1e3:.3f} ms/step  stagger {lanes.stagger*1e3:.3f} ms", flush=True)
    last = lanes.lanes[0]
    lanes.close()
    del lanes
    torch.cuda.empty_cache()
