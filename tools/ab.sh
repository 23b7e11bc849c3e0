#!/bin/bash
# A/B of two builds of the library on the bench workload (run on the GPU box)
for i in 1 2 3; do for lib in ${LIBS:-libpipamd_old.so libpipamd.so}; do
  echo -n "$lib : "
  PIPAMD_LIB=$PWD/piplib_amd/$lib timeout -k 10 120 python3 bench.py --no-cpu --no-dense ${ARGS} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 1
done; done
