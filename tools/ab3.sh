#!/bin/bash
# A/B of builds of the library on the bench workload at several pipeline depths (run on the GPU box)
# LIBS="libpipamd_old.so libpipamd.so" DEPTHS="1 4 12" tools/ab3.sh
for d in ${DEPTHS:-1 4 12}; do for lib in ${LIBS:-libpipamd_old.so libpipamd.so}; do
  st=$((d*8)); [ $st -lt 16 ] && st=16
  echo -n "$lib depth $d ${ARGS}: "
  PIPAMD_LIB=$PWD/piplib_amd/$lib timeout -k 10 180 python3 bench.py --no-cpu --no-dense --pipeline $d --steps $st --warmup $d ${ARGS} 2>gpurun_out/ab3_err.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.1f Mpiv/s  %.3f ms/step  kernel %.3f ms in %d launches' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms'], r['launches_per_step']))" || { tail -5 gpurun_out/ab3_err.log; exit 1; }
done; done
