#!/bin/bash
# Dynamic instruction mix of the device tree kernel on the 1,327-problem parametric set (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_quast
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_quast -- python3 tools/dbg_param.py 1 > gpurun_out/pmc_quast.log 2>&1
tail -1 gpurun_out/pmc_quast.log
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/pmc_quast/*/*counter_collection.csv')[0]
g=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][-40:]
    g[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_INSTS_VALU': n[k]+=1
for k in g:
    if 'quast' in k: print(k, n[k], 'launches', {c[9:]:int(v/n[k]) for c,v in sorted(g[k].items())}, 'per launch')
PY
