#!/bin/bash
# Instruction-cache and wait counters of the pivot kernel, one batch at a time and with 12 in flight (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --list-avail > gpurun_out/pmc_avail.txt 2>&1
for mode in 1 12; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"; do
    tag=$(echo $set | cut -d' ' -f1)_$mode
    rm -rf gpurun_out/pmc_ic/$tag
    timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_ic/$tag -- python3 bench.py --no-cpu --no-dense --no-others --steps 24 --warmup 12 --pipeline $mode > gpurun_out/pmc_ic_$tag.log 2>&1
    echo "$tag rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_ic/*')):
    fs = glob.glob(d + '/*/*counter_collection.csv')
    if not fs: print(d, 'no csv'); continue
    g = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0][:70]
        if 'pip_advance' not in k: continue
        g[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in g: print(d.split('/')[-1], k[-40:], {c: int(v) for c, v in sorted(g[k].items())})
PY
