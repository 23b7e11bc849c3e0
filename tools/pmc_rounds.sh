#!/bin/bash
# instructions per pivot as a function of the round length: the difference is the per-launch
# entry/exit cost (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for r in 100000 48 24 12; do
  rm -rf gpurun_out/pmc_r$r
  ROUND=$r rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/pmc_r$r -- python3 tools/pmc_one.py > gpurun_out/pmc_r$r.log 2>&1
  grep RUN gpurun_out/pmc_r$r.log
  python3 - $r <<'PY'
import csv, glob, collections, sys
f=glob.glob(f'gpurun_out/pmc_r{sys.argv[1]}/*/*counter_collection.csv')[0]
g=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if 'pip_advance_kernel' in r['Kernel_Name']: g[r['Counter_Name']]+=float(r['Counter_Value'])
print("  round", sys.argv[1], {k[3:]:round(v/772044,1) for k,v in sorted(g.items())})
PY
done
