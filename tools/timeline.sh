#!/bin/bash
# kernel timeline of the driver's command (GPU box): rocprofv3 --kernel-trace, then tools/timeline.py prints the last timed region
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --no-cpu --no-dense --no-others --steps ${STEPS:-20} --warmup 5 ${ARGS} > gpurun_out/tl.log 2>&1
python3 tools/timeline.py
