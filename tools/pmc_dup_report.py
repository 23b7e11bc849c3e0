"""Per-piece dynamic instruction counts of the pivot loop from tools/pmc_dup.sh's counter CSVs: the difference between
a build that executes piece n twice (PIP_DUP=n) and the normal build, per pivot of the headline batch."""
import collections
import csv
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {9: "B recycled pivot slot (build + store + publish)", 10: "C flags / next chercher", 11: "exam: apply the flags",
         12: "B row load + pivot-column entry", 13: "A work list", 14: "A pivot row staging + class scan",
         15: "A choisir_piv tournament", 16: "B multipliers (gcd, quotients)", 17: "B row update (products, row gcd, division)",
         18: "B store + publish of a rewritten row", 19: "integrer: search for a non-integral row",
         20: "integrer: cut row (load + fmod)", 21: "integrer: store + publish of the cut"}


def totals(n):
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_dup", f"d{n}", "*", "*counter_collection.csv"))
    if not fs:
        return None
    g = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if "pip_advance_kernel" in r["Kernel_Name"]:
            g[r["Counter_Name"][9:]] += float(r["Counter_Value"])
    return g


def main():
    piv = 772044.0
    base = totals(0)
    print("wave-instructions per pivot, pip_advance_kernel (bulk + tail launches) of one headline batch (%d pivots)" % piv)
    print("%-52s %8s %8s %8s %8s" % ("whole kernel", *[f"{base[k] / piv:.1f}" for k in ("VALU", "SALU", "BRANCH", "LDS")]))
    print("%-52s %8s %8s %8s %8s" % ("piece", "VALU", "SALU", "BRANCH", "LDS"))
    acc = collections.defaultdict(float)
    for n in sorted(NAMES):
        t = totals(n)
        if t is None:
            continue
        d = {k: (t[k] - base[k]) / piv for k in ("VALU", "SALU", "BRANCH", "LDS")}
        for k in d:
            acc[k] += d[k]
        print("%-52s %8.1f %8.1f %8.1f %8.1f" % (f"{n:2d} {NAMES[n]}", d["VALU"], d["SALU"], d["BRANCH"], d["LDS"]))
    print("%-52s %8.1f %8.1f %8.1f %8.1f" % ("sum of the pieces", acc["VALU"], acc["SALU"], acc["BRANCH"], acc["LDS"]))
    print("%-52s %8.1f %8.1f %8.1f %8.1f" % ("rest (entry pass, sort, loop control, epilogue)",
                                              *[base[k] / piv - acc[k] for k in ("VALU", "SALU", "BRANCH", "LDS")]))


if __name__ == "__main__":
    main()
