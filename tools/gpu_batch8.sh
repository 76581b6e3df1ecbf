#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
for v in new old new old; do
unset SLS_HIP_LIB
if [ $v != new ]; then export SLS_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libsls_hip_$v.so; fi
timeout 300 python bench.py --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v','ms/step',round(j['ms_per_step'],1),'acq',round(j['stage_ms_per_step']['acq_gemm'],1))"
done
unset SLS_HIP_LIB
timeout 300 python bench.py --starts 8192 --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8192 starts ms/step',round(j['ms_per_step'],1),'frac',round(j['roofline']['frac'],4))"
