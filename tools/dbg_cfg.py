"""Pipelined throughput of one BASELINE configuration with engine knobs (GPU box).
   python tools/dbg_cfg.py CFG DEPTH [waves] [bulk_min] [round] [round_rows]   CFG: 1 | 4 | 2"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
cfgi, depth = int(sys.argv[1]), int(sys.argv[2])
a = types.SimpleNamespace(waves=int(sys.argv[3]) if len(sys.argv) > 3 else 0,
                          bulk_min=int(sys.argv[4]) if len(sys.argv) > 4 else -1,
                          round=int(sys.argv[5]) if len(sys.argv) > 5 else 0,
                          round_rows=int(sys.argv[6]) if len(sys.argv) > 6 else 0)
cfg = {1: bench.OTHERS[0], 4: bench.OTHERS[1], 2: bench.MAIN}[cfgi]
dev = torch.device("cuda", 0)
lanes = bench.Lanes(cfg, depth, dev, 0, [2000 + 7919 * i for i in range(depth)], a)
def barrier(): torch.cuda.synchronize(dev)
steps = 16 * depth
dt, share = bench.timed(lanes, steps, depth, barrier, -1.0)
tot = lanes.totals(share)
print(f"{cfg['key']} depth {depth} waves {a.waves} bulk_min {a.bulk_min} round {a.round}/{a.round_rows}: "
      f"{tot[0] / dt / 1e6:.1f} Mpiv/s  {dt / steps * 1e3:.3f} ms/step  launches {lanes.lanes[0][0].last_solve_launches()}", flush=True)
