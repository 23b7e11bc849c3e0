#!/bin/bash
# Dynamic instruction count of the pieces of the pivot loop: SQ_INSTS_* of the normal build and
# of the builds that execute one piece twice (tools/build_dups.sh).  GPU box.  Then tools/pmc_dup_report.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_dup
for n in ${DUPS:-0 9 10 11 12 13 14 15 16 17 18 19 20 21}; do
  lib=libpipamd_dup$n.so; [ "$n" = 0 ] && lib=libpipamd.so
  PIPAMD_LIB=$PWD/piplib_amd/$lib timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_dup/d${n:-0} -- python3 tools/pmc_one.py > gpurun_out/pmc_dup_${n:-0}.log 2>&1
  grep RUN gpurun_out/pmc_dup_${n:-0}.log
done
python3 tools/pmc_dup_report.py | tee gpurun_out/pmc_dup_report.txt
