#!/bin/bash
# instruction mix and issue activity of the lean kernel on one un-pipelined solve of the headline batch (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pl_*
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pl_mix -- python3 tools/pmc_one.py > gpurun_out/pl_mix.log 2>&1
python3 tools/pmc_kernels.py gpurun_out/pl_mix
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d gpurun_out/pl_sq -- python3 tools/pmc_one.py > gpurun_out/pl_sq.log 2>&1
python3 tools/pmc_kernels.py gpurun_out/pl_sq
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pl_sq2 -- python3 tools/pmc_one.py > gpurun_out/pl_sq2.log 2>&1
python3 tools/pmc_kernels.py gpurun_out/pl_sq2
tail -1 gpurun_out/pl_mix.log
