#!/bin/bash
# manual tuning sweep: register cap (waves/SIMD) x pivots per round (run on the GPU box)
for lib in libpipamd.so libpipamd_mw6.so libpipamd_mw7.so; do for r in 24 32 48; do
  echo -n "$lib round=$r : "
  PIPAMD_LIB=$PWD/piplib_amd/$lib timeout -k 10 120 python3 bench.py --steps 30 --warmup 4 --no-cpu --no-dense --round $r 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 1
done; done
