"""Prints the kernel timeline of the last timed region of a `tools/timeline.sh` run: per queue, the pivot launches
(name, start and end in ms from the region's first dispatch, workgroups)."""
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/tl/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# regions: gaps > 3 ms without any kernel separate them; take the last region that holds >= 15 lean launches
regions, cur, last_end = [], [], None
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and st - last_end > 3e6:
        regions.append(cur); cur = []
    cur.append(r); last_end = max(last_end or 0, en)
regions.append(cur)
good = [g for g in regions if sum("pip_lean_kernel" in r["Kernel_Name"] for r in g) >= 15]
g = good[-1] if len(good) < 4 else good[-4]  # (the headline's third region when launch_split etc. follow)
t0 = int(g[0]["Start_Timestamp"])
short = lambda n: "lean" if "pip_lean" in n else ("bulk" if "1, 1, false" in n else ("tail" if "pip_advance" in n else ("replayL" if "replay_lanes" in n else ("replay" if "replay" in n else ("load" if "batch_load" in n else ("results" if "results" in n else ("rehouse" if "rehouse" in n else n[:12])))))))
byq = collections.defaultdict(list)
for r in g:
    byq[r["Queue_Id"]].append(r)
print("region: %d dispatches, %.2f ms" % (len(g), (max(int(r["End_Timestamp"]) for r in g) - t0) / 1e6))
for q, rs in sorted(byq.items(), key=lambda kv: int(kv[1][0]["Start_Timestamp"])):
    print("queue", q, " ".join("%s[%.1f-%.1f]" % (short(r["Kernel_Name"]), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6)
                               for r in rs if short(r["Kernel_Name"]) in ("lean", "bulk", "tail", "results", "load")))
