#!/bin/bash
# Dynamic instructions of the one-wave kernel as a function of the pivots per launch: with
# ITER = pivots per launch, every tableau is entered ceil(pivots/ITER) times, so
#   instructions = a * pivots + b * entries + c * tableaux(first entry: full pass + sort)
# (GPU box; prints per-kernel-name SQ_INSTS sums and launch counts)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for it in ${ITERS:-1000000 48 24 12}; do
  rm -rf gpurun_out/pmc_e$it
  ITER=$it WAVES=1 FLAGS=${FLAGS:-1} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_e$it -- python3 tools/pmc_entry.py > gpurun_out/pmc_e$it.log 2>&1
  grep RUN gpurun_out/pmc_e$it.log
  python3 - $it <<'PY'
import csv, glob, collections, sys
f=glob.glob(f'gpurun_out/pmc_e{sys.argv[1]}/*/*counter_collection.csv')[0]
g=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(f)):
    if 'pip_advance_kernel' in r['Kernel_Name']:
        g[r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_INSTS_VALU': n['launches']+=1
print("  iter", sys.argv[1], "launches", n['launches'], {k[9:]:int(v) for k,v in sorted(g.items())})
PY
done
