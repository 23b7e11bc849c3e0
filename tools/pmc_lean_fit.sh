#!/bin/bash
# instruction counts of the lean launch for several pivot budgets (GPU box): per-tableau and per-pivot shares
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for K in ${ROUNDS:-1 8 32 96}; do
  rm -rf gpurun_out/plf_$K
  ROUND=$K rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d gpurun_out/plf_$K -- python3 tools/pmc_lean_fit.py > gpurun_out/plf_$K.log 2>&1
  grep RUN gpurun_out/plf_$K.log
  python3 tools/pmc_kernels.py gpurun_out/plf_$K pip_lean_kernel | grep -E "VALU|SALU|BRANCH|LDS"
done
