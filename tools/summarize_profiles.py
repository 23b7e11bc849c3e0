"""Turns the rocprofv3 output of tools/make_profiles.sh (gpurun_out/prof_*) into profiles/."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def one(pat):
    f = sorted(glob.glob(os.path.join(G, pat)), key=os.path.getmtime)  # merged runs accumulate: newest wins
    assert f, pat
    return f[-1]


shutil.copy(one("prof_np/*/*kernel_stats.csv"), os.path.join(P, f"{tag}_kernel_stats.csv"))
shutil.copy(one("prof_p/*/*kernel_stats.csv"), os.path.join(P, f"{tag}_kernel_stats_pipelined.csv"))


def per_step(path, name):
    steps, cur = [], None
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        if r["Counter_Name"] != name:
            continue
        if "pip_batch_load_kernel" in r["Kernel_Name"]:
            cur = [0.0]
            steps.append(cur)
        elif "pip_advance_kernel" in r["Kernel_Name"] and cur is not None:
            cur[0] += float(r["Counter_Value"])
    return [s[0] for s in steps if s[0] > 0]


fetch = per_step(one("prof_fetch/*/*counter_collection.csv"), "FETCH_SIZE")
write = per_step(one("prof_write/*/*counter_collection.csv"), "WRITE_SIZE")
rd = 2 * 1024 * sum(fetch) / len(fetch)
wr = 1024 * sum(write) / len(write)
old = json.load(open(os.path.join(P, f"{tag}_pmc_hbm.json")))
old.update({"fetch_size_kb_per_step": fetch, "write_size_kb_per_step": write, "hbm_read_bytes_per_step": rd,
            "hbm_write_bytes_per_step": wr, "hbm_bytes_per_step": rd + wr})
json.dump(old, open(os.path.join(P, f"{tag}_pmc_hbm.json"), "w"), indent=1)
print("read GB", rd / 1e9, "write GB", wr / 1e9, "steps", len(fetch), len(write))
