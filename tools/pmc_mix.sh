#!/bin/bash
# Dynamic instruction mix of one un-pipelined headline step (GPU box): per-kernel SQ_INSTS_* sums
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_mix
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_mix -- python3 tools/pmc_one.py > gpurun_out/pmc_mix.log 2>&1
grep RUN gpurun_out/pmc_mix.log
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/pmc_mix/*/*counter_collection.csv')[0]
g=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][:60]
    g[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_INSTS_VALU': n[k]+=1
for k in g:
    if 'pip_' in k: print(k, n[k], {c[9:]:int(v) for c,v in sorted(g[k].items())})
PY
