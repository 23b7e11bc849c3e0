#!/bin/bash
# bench.py's stagger (lanes start a fraction of a step apart) on long and short runs (GPU box)
for args in "--steps 96 --warmup 24" "--steps 20 --warmup 5" "--steps 40 --warmup 10"; do for st in "" "--stagger 0"; do
  echo -n "$args $st: "
  timeout -k 10 180 python3 bench.py --no-cpu --no-dense --no-others $args $st 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Mpiv/s  %.3f ms/step depth %d' % (d['value']/1e6, d['ms_per_step'], d['config']['pipeline_depth']))"
done; done
