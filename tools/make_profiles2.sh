#!/bin/bash
# Second part of profiles/: phase profile + event counts (diagnostic builds must exist:
# python -m piplib_amd.build --profile; --profile-events) and the SQ issue counters.  GPU box.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/dbg_prof.py 10000 2>/dev/null | tail -17 > gpurun_out/phase_bulk.txt
python3 tools/dbg_prof.py 64 1 2>/dev/null | tail -17 > gpurun_out/phase_lone.txt
EVENTS=1 python3 tools/dbg_prof.py 10000 2>/dev/null | tail -37 | head -20 > gpurun_out/phase_events.txt
rm -rf gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-dense --pipeline 1 > gpurun_out/pmc_sq.log 2>&1
bash tools/pmc_base.sh
