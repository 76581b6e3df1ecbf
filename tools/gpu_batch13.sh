#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -q -x --durations=4 > gpurun_out/r02/pytest_lbfgs4.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02/pytest_lbfgs4.log; tail -6 gpurun_out/r02/pytest_lbfgs4.log
for st in 65536 8192; do timeout 300 python bench.py --starts $st --no-cpu-baseline --steps 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$st starts', round(j['ms_per_step'],1), round(j['roofline']['frac'],4), 'lbfgs', round(j['stage_ms_per_step']['lbfgs'],1), 'value', round(j['value']), j['result'] if 'result' in j else '')"; done
