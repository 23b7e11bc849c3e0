import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gmpfix import gmp_fixture
import pipbatch as pb
from piplib_amd import engine as eng
e = eng.Engine(0)
for seed in (81, 82, 83, 84):
    probs, flags, recs, sha = gmp_fixture("param%d" % seed)
    keep = [(p, r) for p, r in zip(probs, recs) if "status" in r and not r["wrap128"] and r["pivots"] <= 20000]
    many = eng.solve_tableaux_lockstep128(e, [p for p, _ in keep])
    served, back = e.last_device_tree()
    ok = sum(1 for (p, r), (t, rc, st, piv) in zip(keep, many) if rc == 0 and piv == r["pivots"] and pb.squash(t) == pb.squash("void\n" if r["status"] == pb.ST_VOID else r["text"]))
    print("param%d: %d problems, device tree served %d, handed back %d; equal to the GMP record: %d; rc!=0: %d" % (seed, len(keep), served, back, ok, sum(1 for m in many if m[1] != 0)))
