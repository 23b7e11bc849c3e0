"""Turns the output of tools/make_profiles2.sh (gpurun_out/) into profiles/r01_phase_profile.txt
and profiles/r01_pmc_sq_issue.json."""
import csv, collections, glob, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
rd = lambda n: open(os.path.join(G, n)).read().rstrip("\n").splitlines()
out = ["# tools/dbg_prof.py, final kernel of round 1: (1) 10,000 tableaux, default waves (bulk regime, 3 launches); "
       "(2) 64 tableaux, 1 wave each (lone regime); (3) event counts per pivot"]
out += rd("phase_bulk.txt") + rd("phase_lone.txt")
out += ["# event counts per pivot (python -m piplib_amd.build --profile-events; its kernel time is meaningless)"] + rd("phase_events.txt")[1:]
open(os.path.join(P, "r01_phase_profile.txt"), "w").write("\n".join(out) + "\n")
f = sorted(glob.glob(os.path.join(G, "pmc_sq/*/*counter_collection.csv")), key=os.path.getmtime)[-1]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "advance" not in r["Kernel_Name"]:
        continue
    a = agg.setdefault(int(r["Dispatch_Id"]), {"wg": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"]),
                                               "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                                               "c": collections.defaultdict(float)})
    a["c"][r["Counter_Name"]] += float(r["Counter_Value"])
step = list(agg.items())[3:6]
launches = []
for k, a in step:
    c, wc, cyc = a["c"], a["c"]["SQ_WAVE_CYCLES"], a["ms"] * 1e-3 * 2.4e9
    launches.append({"dispatch": k, "workgroup_size": a["wg"], "workgroups": a["grid"] // a["wg"], "ms": round(a["ms"], 3),
                     "SQ_WAVE_CYCLES_quad": wc,
                     "share_of_wave_cycles": {n[3:]: round(v / wc, 3) for n, v in c.items() if n != "SQ_WAVE_CYCLES"},
                     "avg_resident_waves_per_CU": round(wc * 4 / cyc / 256, 1),
                     "valu_issue_utilisation_per_SIMD": round(c["SQ_ACTIVE_INST_VALU"] * 4 / cyc / 1024, 2),
                     "salu_issue_utilisation_per_SIMD": round(c["SQ_ACTIVE_INST_SCA"] * 4 / cyc / 1024, 2)})
doc = json.load(open(os.path.join(P, "r01_pmc_sq_issue.json")))
doc["launches_of_one_step"] = launches
doc["vector_issue_ms_per_SIMD_per_step"] = round(sum(a["c"]["SQ_ACTIVE_INST_VALU"] for _, a in step) * 4 / 1024 / 2.4e9 * 1e3, 2)
json.dump(doc, open(os.path.join(P, "r01_pmc_sq_issue.json"), "w"), indent=1)
for o in launches:
    print(o["ms"], o["avg_resident_waves_per_CU"], o["valu_issue_utilisation_per_SIMD"], o["salu_issue_utilisation_per_SIMD"], o["share_of_wave_cycles"])
print("vector issue ms/SIMD/step", doc["vector_issue_ms_per_SIMD_per_step"])
