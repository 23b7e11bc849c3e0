#!/bin/bash
# Regenerates the raw material of profiles/ on the GPU box (run through gpurun from the repo
# root); tools/summarize_profiles.py then turns gpurun_out/prof_* into the files under profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# --tail-waves 4: the engine of the default run (12 batches in flight), whose launches bench.py's roofline times
B="python3 bench.py --no-cpu --no-dense --no-others --tail-waves 4"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_np -- $B --steps 5 --warmup 1 --pipeline 1 > gpurun_out/prof_np.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_p -- $B > gpurun_out/prof_p.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- $B --steps 2 --warmup 1 --pipeline 1 > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- $B --steps 2 --warmup 1 --pipeline 1 > gpurun_out/prof_write.log 2>&1
tail -1 gpurun_out/prof_np.log | cut -c1-300
# the same kernel with row skipping off (roofline_dense_mode): one un-pipelined run that keeps the dense leg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dense -- python3 bench.py --no-cpu --no-others --tail-waves 4 --steps 2 --warmup 1 --pipeline 1 > gpurun_out/prof_dense.log 2>&1
