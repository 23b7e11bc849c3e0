"""Per-kernel counter sums of a rocprofv3 --pmc run: python3 tools/pmc_kernels.py <dir> [kernel substring ...]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
want = sys.argv[2:] or ["pip_lean_kernel", "pip_advance_kernel"]
f = sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    for w in want:
        if w in r["Kernel_Name"]:
            key = w + (" x%d" % int(r["Workgroup_Size"]) if "Workgroup_Size" in r else "")
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[key].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(k, "dispatches", len(nd[k]))
    for c in sorted(acc[k]):
        print("   %-24s %16.0f  per dispatch %14.1f" % (c, acc[k][c], acc[k][c] / len(nd[k])))
