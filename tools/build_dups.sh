#!/bin/bash
# builds libpipamd_dup<n>.so for the pieces of the pivot loop listed in DUPS (see PIP_DUP in csrc/pip_advance.h)
set -e
cd "$(dirname "$0")/.."
for n in ${DUPS:-9 10 11 12 13 14 15 16 17 18 19 20 21}; do
  PIP_DUP=$n python3 -m piplib_amd.build > /tmp/build_dup$n.log 2>&1 || { echo "dup $n failed"; tail -5 /tmp/build_dup$n.log; exit 1; }
  echo "built dup $n"
done
