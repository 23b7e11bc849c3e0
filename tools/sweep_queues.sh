#!/bin/bash
# manual tuning sweep: HIP hardware queues x batches in flight (run on the GPU box)
for q in 16 24; do for p in 8 12 16 20; do
  echo -n "GPU_MAX_HW_QUEUES=$q pipeline=$p : "
  GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python3 bench.py --steps 80 --warmup 20 --no-cpu --no-dense --pipeline $p 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
done; done
