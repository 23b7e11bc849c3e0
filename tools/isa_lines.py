"""Static instruction counts per source line of one kernel from a `-gline-tables-only --save-temps` .s
(diagnostics): python tools/isa_lines.py kernel.s [lo hi]  -> per-line valu/salu/lds/vmem/spill counts."""
import re, sys, collections
cnt = collections.defaultdict(lambda: collections.Counter())
cur = 0
fileno = None
for ln in open(sys.argv[1]):
    t = ln.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = int(m.group(2)) if True else cur
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    c = cnt[cur]
    if op.startswith("v_writelane") or op.startswith("v_readlane"):
        c["lane"] += 1
    if op.startswith("v_"): c["valu"] += 1
    elif op.startswith("s_waitcnt"): c["wait"] += 1
    elif op.startswith("s_"): c["salu"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): c["vmem"] += 1
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
tot = collections.Counter()
for l in sorted(cnt):
    if lo <= l <= hi:
        c = cnt[l]; tot.update(c)
        if len(sys.argv) > 2:
            print(l, dict(c))
print("total", dict(tot))
