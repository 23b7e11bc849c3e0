#!/bin/bash
# batches in flight on the driver's command (20 steps, 5 warm-up) and at 96 steps (GPU box): DEPTHS="7 10 14" tools/sweep_depth.sh
for p in ${DEPTHS:-7 10 12 14 16 20}; do for st in "20 5" "96 12"; do
  set -- $st
  echo -n "pipeline=$p steps=$1 : "
  timeout -k 10 150 python3 bench.py --steps $1 --warmup $2 --no-cpu --no-dense --no-others --pipeline $p ${ARGS} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f M  regions %s' % (d['value']/1e6, d['regions_ms']))" || exit 1
done; done
