#!/bin/bash
# robustness of the bench line against the driver's --steps/--warmup choice (run on the GPU box)
for a in "--steps 5 --warmup 1" "--steps 10 --warmup 2" "--steps 20 --warmup 5" "--steps 48 --warmup 12" "--steps 100 --warmup 10" "--steps 10 --warmup 2 --pipeline 5"; do
  echo -n "$a : "
  timeout -k 10 120 python3 bench.py $a --no-cpu --no-dense 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['pipeline_depth'])" || exit 1
done
