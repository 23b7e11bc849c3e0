"""Kernel timeline of lone steps (`ARGS="--pipeline 1" STEPS=6 tools/timeline.sh` writes the trace): every dispatch of the last
three steps with start / end in ms and the gap to its predecessor."""
import csv, glob
f = sorted(glob.glob("gpurun_out/tl/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: "lean" if "pip_lean" in n else ("bulk" if "1, 1, false" in n else ("tail" if "pip_advance" in n else ("replayL" if "replay_lanes" in n else ("replay" if "replay" in n else ("load" if "batch_load" in n else ("results" if "results" in n else ("rehouse" if "rehouse" in n else n[:14])))))))
loads = [i for i, r in enumerate(rows) if "batch_load" in r["Kernel_Name"]]
lo = loads[-8] if len(loads) >= 8 else loads[0]
hi = loads[-4]
t0 = int(rows[lo]["Start_Timestamp"]); prev = t0
for r in rows[lo:hi]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-10s start %8.3f  dur %7.3f  gap %7.3f" % (short(r["Kernel_Name"]), (st - t0) / 1e6, (en - st) / 1e6, (st - prev) / 1e6))
    prev = en
