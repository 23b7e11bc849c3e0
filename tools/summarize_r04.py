"""Turns the rocprofv3 output of tools/make_profiles_r04.sh (gpurun_out/p4_*) into profiles/r04_*."""
import collections, csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
CLK = 2.4e9  # MI355X engine clock used for the utilisation figures


def one(pat):
    f = sorted(glob.glob(os.path.join(G, pat)), key=os.path.getmtime)
    assert f, pat
    return f[-1]


for src, dst in (("p4_np", "r04_kernel_stats.csv"), ("p4_p", "r04_kernel_stats_pipelined.csv"),
                 ("p4_dense", "r04_dense_mode_kernel_stats.csv"), ("p4_cfg4", "r04_cfg4_kernel_stats.csv")):
    shutil.copy(one(src + "/*/*kernel_stats.csv"), os.path.join(P, dst))


PIVOT_KERNELS = ("pip_lean_kernel", "pip_advance_kernel")  # the launches that pivot: the lean bulk kernel and the general one


def per_step(path, name, kernel=PIVOT_KERNELS, pick=None):
    """counter `name` of `kernel` summed per step (a step starts with a pip_batch_load_kernel dispatch)"""
    steps, cur = [], None
    for r in sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"])):
        if r["Counter_Name"] != name:
            continue
        if "pip_batch_load_kernel" in r["Kernel_Name"]:
            cur = [0.0, 0.0]
            steps.append(cur)
        elif any(k in r["Kernel_Name"] for k in kernel) and cur is not None and (pick is None or pick(r)):
            cur[0] += float(r["Counter_Value"])
            cur[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return [s for s in steps if s[0] > 0]


def hbm(tag):
    f = per_step(one(f"p4_{tag}fetch/*/*counter_collection.csv"), "FETCH_SIZE")
    w = per_step(one(f"p4_{tag}write/*/*counter_collection.csv"), "WRITE_SIZE")
    return f, w


out = {"correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B-per-lane coalesced reads, MI355X_MICROARCH.md "
                     "HBM section); WRITE_SIZE as read; both in KiB",
       "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --no-cpu "
                  "--no-dense --no-others --tail-waves 4 --steps 2 --warmup 1 --pipeline 1   (dense: without --no-dense; "
                  "cfg4: python3 tools/cfg_rate.py 4 1)", "batch_per_gpu": 10000}
f, w = hbm("")
# steps of the headline batch (the per-batch counter passes of Lanes.__init__ and the warm-up come first: all are the
# same batch, the mean over them is the step)
rd, wr = 2 * 1024 * sum(s[0] for s in f) / len(f), 1024 * sum(s[0] for s in w) / len(w)
out.update({"hbm_read_bytes_per_step": rd, "hbm_write_bytes_per_step": wr, "hbm_bytes_per_step": rd + wr, "steps_seen": [len(f), len(w)]})
# the lean launch alone (the dominant kernel of bench.py's `roofline`)
fl = per_step(one("p4_fetch/*/*counter_collection.csv"), "FETCH_SIZE", kernel=("pip_lean_kernel",))
wl = per_step(one("p4_write/*/*counter_collection.csv"), "WRITE_SIZE", kernel=("pip_lean_kernel",))
if fl and wl:
    rdl, wrl = 2 * 1024 * sum(s[0] for s in fl) / len(fl), 1024 * sum(s[0] for s in wl) / len(wl)
    out["lean_launch"] = {"hbm_read_bytes": rdl, "hbm_write_bytes": wrl, "hbm_bytes": rdl + wrl,
                          "kernel_ms_under_the_counter_pass": sum(s[1] for s in fl) / len(fl)}
# dense mode: the NOSKIP launches are the long four-wave ones (>= 8 ms)
fd = per_step(one("p4_dense_fetch/*/*counter_collection.csv"), "FETCH_SIZE")
wd = per_step(one("p4_dense_write/*/*counter_collection.csv"), "WRITE_SIZE")
big = lambda steps: [s for s in steps if s[1] > 8.0]
if big(fd) and big(wd):
    rdd, wrd = 2 * 1024 * sum(s[0] for s in big(fd)) / len(big(fd)), 1024 * sum(s[0] for s in big(wd)) / len(big(wd))
    msd = sum(s[1] for s in big(fd)) / len(big(fd))
    out["dense_mode"] = {"hbm_read_bytes_per_step": rdd, "hbm_write_bytes_per_step": wrd, "hbm_bytes_per_step": rdd + wrd,
                         "kernel_ms_under_the_counter_pass": msd, "hbm_GBps": (rdd + wrd) / msd / 1e6,
                         "note": "PIPAMD_T_NOSKIP launches only (the steps whose pivot kernels ran for more than 8 ms)"}
try:
    fc, wc = hbm("cfg4_")
    rdc, wrc = 2 * 1024 * sum(s[0] for s in fc) / len(fc), 1024 * sum(s[0] for s in wc) / len(wc)
    out["cfg4"] = {"hbm_read_bytes_per_step": rdc, "hbm_write_bytes_per_step": wrc, "hbm_bytes_per_step": rdc + wrc,
                   "kernel_ms_per_step": sum(s[1] for s in fc) / len(fc)}
except AssertionError:
    pass
json.dump(out, open(os.path.join(P, "r04_pmc_hbm.json"), "w"), indent=1)

# instruction mix of one un-pipelined headline step
mix = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(one("p4_mix/*/*counter_collection.csv"))):
    mix[r["Kernel_Name"].split("(")[0]][r["Counter_Name"][9:]] += float(r["Counter_Value"])
piv = 772044.0
piv_lean = 772044.0  # of them in the two lean launches (the second resumes what the first paused): all of them on this batch
lines = ["# tools/make_profiles_r04.sh step 4: rocprofv3 --pmc SQ_INSTS_* -- python3 tools/pmc_one.py",
         "# one un-pipelined step of the headline batch (10,000 tableaux, 772,044 pivots): wave-instructions per kernel"]
tot, tot_lean = collections.defaultdict(float), collections.defaultdict(float)
for k, c in mix.items():
    if "pip_" in k:
        lines.append(f"{k} {dict((n, int(v)) for n, v in sorted(c.items()))}")
    if any(pk in k for pk in PIVOT_KERNELS):
        for n, v in c.items():
            tot[n] += v
            if "pip_lean_kernel" in k:
                tot_lean[n] += v
lines.append("pip_lean_kernel (both launches) per pivot of its own (772,044): " + ", ".join(f"{n} {v / piv_lean:.1f}" for n, v in sorted(tot_lean.items())))
lines.append("all pivot launches per pivot (772,044): " + ", ".join(f"{n} {v / piv:.1f}" for n, v in sorted(tot.items())))
open(os.path.join(P, "r04_pmc_inst_mix.txt"), "w").write("\n".join(lines) + "\n")

# issue utilisation of the launches (counter passes serialise the kernels: these are lone launches whatever --pipeline)
issue = {"lean_kernel_wave_instructions_per_pivot": {n: round(v / piv_lean, 1) for n, v in sorted(tot_lean.items())},
         "lean_kernel_wave_instructions_per_pivot_total": round(sum(tot_lean[n] for n in ("VALU", "SALU", "BRANCH", "LDS", "VMEM_RD", "VMEM_WR", "SMEM")) / piv_lean, 1),
         "wave_instructions_per_pivot": {n: round(v / piv, 1) for n, v in sorted(tot.items())},
         "wave_instructions_per_pivot_total": round(sum(tot[n] for n in ("VALU", "SALU", "BRANCH", "LDS", "VMEM_RD", "VMEM_WR", "SMEM")) / piv, 1)}
for mode in (1, 12):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for pat in (f"p4_sq_{mode}/*/*counter_collection.csv", f"p4_sq2_{mode}/*/*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(one(pat))):
            if not any(pk in r["Kernel_Name"] for pk in PIVOT_KERNELS):
                continue
            k = "lean" if "pip_lean_kernel" in r["Kernel_Name"] else ("bulk" if "1, 1, false" in r["Kernel_Name"] else "tail")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (pat, r["Dispatch_Id"]) not in seen and pat.startswith(f"p4_sq_{mode}"):
                seen.add((pat, r["Dispatch_Id"]))
                agg[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                agg[k]["_launches"] += 1
    for k, c in agg.items():
        cyc = c["_ns"] * 1e-9 * CLK
        wc = c["SQ_WAVE_CYCLES"]
        issue[f"{k}_pipeline{mode}"] = {
            "launches": int(c["_launches"]), "avg_launch_ms": c["_ns"] / max(1, c["_launches"]) / 1e6,
            "share_of_wave_cycles": {n[3:]: round(c[n] / wc, 3) for n in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA",
                                                                           "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_ANY",
                                                                           "SQ_WAIT_INST_ANY") if n in c},
            "avg_resident_waves_per_CU": round(wc * 4 / cyc / 256, 1),  # of 32 (lean: 8 per SIMD) / 24 (general one-wave kernel)
            "valu_issue_utilisation_per_SIMD": round(c["SQ_ACTIVE_INST_VALU"] * 4 / cyc / 1024, 3),
            "salu_issue_utilisation_per_SIMD": round(c["SQ_ACTIVE_INST_SCA"] * 4 / cyc / 1024, 3),
            "icache_miss_rate": round(c["SQC_ICACHE_MISSES"] / max(1.0, c["SQC_ICACHE_REQ"]), 6)}
issue["note"] = ("SQ_* cycle counters are quad-cycles (MI355X_MICROARCH.md); utilisation assumes 2.4 GHz.  A counter pass serialises the "
                 "kernels, so pipeline12 shows the same lone launches as pipeline1 (12 batches, each launch on its own).  lean = pip_lean_kernel, bulk = the general one-wave launch over what it left, tail = the four-wave launches.")
json.dump(issue, open(os.path.join(P, "r04_pmc_issue.json"), "w"), indent=1)

print(json.dumps({k: v for k, v in out.items() if "bytes" in k or k in ("dense_mode", "cfg4")}, indent=1))
print(json.dumps(issue, indent=1)[:1500])
