#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -4
for st in 8192 65536; do
timeout 300 python bench.py --starts $st --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('starts',$st,'ms/step',round(j['ms_per_step'],1),'frac',round(j['roofline']['frac'],3)); print('  ',{k:round(v,1) for k,v in j['stage_ms_per_step'].items()})"
done
