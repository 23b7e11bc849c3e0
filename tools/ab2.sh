#!/bin/bash
# A/B of library builds x bench arguments (GPU box): LIBS="a.so b.so" ARGSETS="--round 32|--round 48"
IFS='|' read -ra SETS <<< "${ARGSETS:-}"
[ ${#SETS[@]} -eq 0 ] && SETS=("")
for i in 1 2; do for lib in ${LIBS:-libpipamd.so}; do for a in "${SETS[@]}"; do
  echo -n "$lib $a : "
  PIPAMD_LIB=$PWD/piplib_amd/$lib timeout -k 10 120 python3 bench.py --no-cpu --no-dense $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 1
done; done; done
