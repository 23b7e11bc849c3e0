#!/bin/bash
# manual tuning sweep: pivots per round with 12 batches in flight (run on the GPU box)
for r in 24 32 48 64 96 128; do
  echo -n "round=$r : "
  timeout -k 10 120 python3 bench.py --no-cpu --no-dense --round $r 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches_per_step'])" || exit 1
done
