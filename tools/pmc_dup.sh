#!/bin/bash
# Dynamic instruction count of the pieces of the pivot loop: SQ_INSTS_* of the normal build and
# of the builds that execute one piece twice (PIP_DUP=n python -m piplib_amd.build).  GPU box.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in ${DUPS:-0 9 10 11 12}; do
  lib=libpipamd_dup$n.so; [ "$n" = 0 ] && lib=libpipamd.so
  PIPAMD_LIB=$PWD/piplib_amd/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d gpurun_out/pmc_dup/d${n:-0} -- python3 tools/pmc_one.py > gpurun_out/pmc_dup_${n:-0}.log 2>&1
  grep RUN gpurun_out/pmc_dup_${n:-0}.log
done
