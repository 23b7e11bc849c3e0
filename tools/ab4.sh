#!/bin/bash
# A/B of builds of the library on the headline workload (GPU box): every library in turn, REPS times over, so that
# drift of the box shows as spread within a library rather than as a difference between them.
# LIBS="libpipamd_old.so libpipamd.so" REPS=3 STEPS=96 tools/ab4.sh
for rep in $(seq 1 ${REPS:-3}); do for lib in ${LIBS:-libpipamd_old.so libpipamd.so}; do
  echo -n "$lib rep $rep: "
  PIPAMD_LIB=$PWD/piplib_amd/$lib timeout -k 10 180 python3 bench.py --no-cpu --no-dense --no-others --steps ${STEPS:-96} --warmup 12 ${ARGS} 2>gpurun_out/ab4_err.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.1f Mpiv/s  regions %s  kernel %.3f ms %s' % (d['value']/1e6, d['regions_ms'], r['kernel_ms'], [round(l['ms'],3) for l in (r.get('launches') or [])]))" || { tail -5 gpurun_out/ab4_err.log; exit 1; }
done; done
