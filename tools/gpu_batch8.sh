#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "maxim or compaction" 2>&1 | tail -2
for st in 8192 65536; do
timeout 300 python bench.py --starts $st --no-cpu-baseline --steps 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('starts',$st,'ms/step',round(j['ms_per_step'],1),'frac',round(j['roofline']['frac'],4),'lbfgs',round(j['stage_ms_per_step']['lbfgs'],1), j['result']['best_value'])"
done
