#!/bin/bash
# A/B of acq_gemm's scheduling forms through bench.py itself: step time, fraction of the fp64 MFMA peak and the measured fabric traffic.
IFS=$'\n'; for cfg in $(printf "%s\n" "SLS_ACQ_WG_PER_CU=1" "SLS_PERSIST=1 SLS_GATE_PHASE=0" "SLS_PERSIST=1 SLS_GATE_PHASE=0 SLS_GATE_EVERY=2" "SLS_PERSIST=1 SLS_GATE_PHASE=0 SLS_GATE_EVERY=4" "SLS_PERSIST=1 SLS_GATE_PHASE=0 SLS_GATE_EVERY=8" "SLS_ACQ_WG_PER_CU=2 SLS_GATE_PHASE=2000"); do IFS=" "
  echo "== $cfg"
  env $cfg python bench.py --steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; d=r['traffic_detail']
        print('ms_per_step %.1f frac %.4f traffic %.1f GB (x%.2f algorithmic) launch ms under profiler %.2f' % (j['ms_per_step'], r['frac'], (r['traffic'] or 0)/1e9, d.get('traffic_over_algorithmic',0), d.get('avg_launch_ms_under_profiler',0)))
"
done
