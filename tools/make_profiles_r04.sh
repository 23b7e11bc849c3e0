#!/bin/bash
# Round 4: regenerates the raw material of profiles/r04_* on the GPU box (through gpurun from the repo root);
# tools/summarize_r04.py then turns gpurun_out/p4_* into the files under profiles/.
# Counter passes and trace passes are separate runs (never --pmc together with a trace domain).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/p4_*
B="python3 bench.py --no-cpu --no-dense --no-others --tail-waves 4 --lone 0"
# 1. kernel stats: one batch at a time (the isolated launches bench.py's roofline times; --lone 0 keeps the three-launch
#    sequence of the pipelined run) and the default 16 in flight
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_np -- $B --steps 5 --warmup 1 --pipeline 1 > gpurun_out/p4_np.log 2>&1
echo "np done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_p -- $B --steps 20 --warmup 5 > gpurun_out/p4_p.log 2>&1
echo "p done"
# 2. HBM traffic of the pivot kernel (FETCH_SIZE / WRITE_SIZE, own passes)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p4_fetch -- $B --steps 2 --warmup 1 --pipeline 1 > gpurun_out/p4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p4_write -- $B --steps 2 --warmup 1 --pipeline 1 > gpurun_out/p4_write.log 2>&1
echo "hbm done"
# 3. the same with row skipping off (roofline_dense_mode): trace pass, then the two counter passes
D="python3 bench.py --no-cpu --no-others --tail-waves 4 --lone 0 --steps 2 --warmup 1 --pipeline 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_dense -- $D > gpurun_out/p4_dense.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p4_dense_fetch -- $D > gpurun_out/p4_dense_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p4_dense_write -- $D > gpurun_out/p4_dense_write.log 2>&1
echo "dense done"
# 4. issue counters of the pivot kernel: instruction mix, then activity / waits (one batch at a time and 12 in flight)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d gpurun_out/p4_mix -- python3 tools/pmc_one.py > gpurun_out/p4_mix.log 2>&1
for mode in 1 12; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d gpurun_out/p4_sq_$mode -- $B --steps 24 --warmup 12 --pipeline $mode > gpurun_out/p4_sq_$mode.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/p4_sq2_$mode -- $B --steps 24 --warmup 12 --pipeline $mode > gpurun_out/p4_sq2_$mode.log 2>&1
done
echo "sq done"
# 5. configs[4] (128-bit entries): kernel stats and HBM traffic
C="python3 tools/cfg_rate.py 4"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_cfg4 -- $C > gpurun_out/p4_cfg4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p4_cfg4_fetch -- $C 1 > gpurun_out/p4_cfg4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p4_cfg4_write -- $C 1 > gpurun_out/p4_cfg4_write.log 2>&1
echo "cfg4 done"
echo "all done"
