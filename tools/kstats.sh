#!/bin/bash
# per-kernel time of the pipelined headline run (GPU box): rocprofv3 --kernel-trace --stats, printed as a table
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --no-cpu --no-dense --no-others --steps ${STEPS:-48} --warmup 12 ${ARGS} > gpurun_out/ks.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/ks/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    print("%-90s calls %6s total %10.3f ms avg %9.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 gpurun_out/ks.log | cut -c1-300
