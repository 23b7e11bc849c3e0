#!/bin/bash
# SQ_INSTS_* of one un-pipelined solve with the normal build (GPU box); prints instructions per pivot
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_base
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_base -- python3 tools/pmc_one.py > gpurun_out/pmc_base.log 2>&1
grep RUN gpurun_out/pmc_base.log
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/pmc_base/*/*counter_collection.csv')[0]
g=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if 'pip_advance_kernel' in r['Kernel_Name']: g[r['Counter_Name']]+=float(r['Counter_Value'])
print({k[9:]:round(v/772044,1) for k,v in sorted(g.items())})
PY
