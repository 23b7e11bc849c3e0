#!/bin/bash
# Average latency of the pivot kernel's memory / LDS instructions (accumulated in-flight levels / instruction counts), GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in 1 12; do
  for set in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVES SQ_INSTS_SALU"; do
    tag=$(echo $set | cut -d' ' -f1)_$mode
    rm -rf gpurun_out/pmc_lat/$tag
    timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_lat/$tag -- python3 bench.py --no-cpu --no-dense --no-others --steps 24 --warmup 12 --pipeline $mode > gpurun_out/pmc_lat_$tag.log 2>&1
    echo "$tag rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_lat/*')):
    fs = glob.glob(d + '/*/*counter_collection.csv')
    if not fs: print(d, 'no csv'); continue
    g = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0][:70]
        if 'pip_advance' not in k: continue
        g[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in g: print(d.split('/')[-1], k[-40:], {c: int(v) for c, v in sorted(g[k].items())})
PY
