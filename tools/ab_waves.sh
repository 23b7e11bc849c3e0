#!/bin/bash
# tail launch with 1 or 4 waves per tableau, one batch at a time and 12 in flight (GPU box)
for a in "--pipeline 1 --steps 6 --warmup 2" "--pipeline 1 --steps 6 --warmup 2 --waves 1" ""  "--waves 1"; do
  echo -n "$a : "
  timeout -k 10 120 python3 bench.py --no-cpu --no-dense $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches_per_step'])" || exit 1
done
