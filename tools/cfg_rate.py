"""Pipelined and lone rates of one of bench.py's other configurations (GPU box): python3 tools/cfg_rate.py 1|4 [lanes [waves per tableau [bulk pivot budget [bulk spare rows]]]]"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
oc = [c for c in bench.OTHERS if c["key"] == "configs[%s]" % sys.argv[1]][0]
od = int(sys.argv[2]) if len(sys.argv) > 2 else 12
args = types.SimpleNamespace(waves=int(sys.argv[3]) if len(sys.argv) > 3 else 0, round=int(sys.argv[4]) if len(sys.argv) > 4 else 0, round_rows=int(sys.argv[5]) if len(sys.argv) > 5 else 0, tail_waves=0, blocking_wait=-1, bulk_min=int(sys.argv[6]) if len(sys.argv) > 6 else -1, copy_rows=False)
dev = torch.device("cuda", 0)
def barrier(): torch.cuda.synchronize(dev)
fuse = int(os.environ.get("CFG_FUSE", "0")) or (max(1, min(16, 10000 // oc["batch"])) if oc["ebits"] == 64 else 1)
ol = bench.Lanes(oc, od, dev, 0, [[0] for _ in range(od)] if oc.get("family") else [[i * fuse + k for k in range(fuse)] for i in range(od)], args, fuse=fuse)
regs = sorted((bench.timed(ol, (2 if oc.get("family") else 8) * od * fuse, od, barrier) for _ in range(3)), key=lambda r: r[0])
dt, sh = regs[1]
t = ol.totals(sh)
o1 = bench.Lanes(oc, 1, dev, 0, [[0]], args)
n1 = int(os.environ.get("LONE_STEPS", "2" if oc.get("family") else "12"))
dt1, sh1 = bench.timed(o1, n1, 2, barrier)
t1 = o1.totals(sh1)
print("round %d rows %d: " % (args.round, args.round_rows), end="")
print("%s lib %s: %.1f M pivots/s with %d lanes (fuse %d), %.2f M one batch at a time; %s" % (oc["key"], os.path.basename(bench.__dict__.get("x", "") or os.environ.get("PIPAMD_LIB", "libpipamd.so")), t[0] / dt / 1e6, od, fuse, t1[0] / dt1 / 1e6, ol.status_histogram()))
