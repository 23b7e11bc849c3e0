#!/bin/bash
# PC sampling of one un-pipelined headline solve (GPU box): where the waves of the pivot kernel spend their time.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pcs
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout -k 5 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${PCS_METHOD:-stochastic} --pc-sampling-unit ${PCS_UNIT:-cycles} --pc-sampling-interval ${PCS_INTERVAL:-1048576} --kernel-trace --output-format csv -d gpurun_out/pcs -- python3 tools/pmc_one.py > gpurun_out/pcs.log 2>&1
echo "rc=$?" >> gpurun_out/pcs.log
ls -la gpurun_out/pcs/*/ >> gpurun_out/pcs.log 2>&1
tail -5 gpurun_out/pcs.log
