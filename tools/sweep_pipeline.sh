#!/bin/bash
# manual tuning sweep: batches in flight x stagger (run on the GPU box)
for p in 2 3 4 6; do for s in 0 -1 1.5 3; do
  echo -n "pipeline=$p stagger=$s : "
  timeout -k 10 120 python3 bench.py --steps 30 --warmup 4 --no-cpu --no-dense --pipeline $p --stagger $s 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
done; done
