#!/bin/bash
# A/B of one library under two environments on the headline workload (GPU box), turn by turn:
# A="PIPAMD_NO_LEAN=1" B="PIPAMD_NO_LEAN=0" REPS=3 STEPS=96 tools/ab_env.sh
for rep in $(seq 1 ${REPS:-3}); do for v in "${A:-PIPAMD_NO_LEAN=1}" "${B:-PIPAMD_NO_LEAN=0}"; do
  echo -n "$v rep $rep: "
  env $v timeout -k 10 180 python3 bench.py --no-cpu --no-dense --no-others --steps ${STEPS:-96} --warmup 12 ${ARGS} 2>gpurun_out/ab_env_err.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.1f Mpiv/s  regions %s  kernel %.3f ms %s fin %s' % (d['value']/1e6, d['regions_ms'], r['kernel_ms'], [round(l['ms'],3) for l in (r.get('launches') or [])], d.get('finished_fraction')))" || { tail -5 gpurun_out/ab_env_err.log; exit 1; }
done; done
