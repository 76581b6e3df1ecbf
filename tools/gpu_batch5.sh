#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=5 > gpurun_out/r02/pytest_b5.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02/pytest_b5.log; tail -12 gpurun_out/r02/pytest_b5.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02/bench3.json 2> gpurun_out/r02/bench3.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r02/bench3.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "value", j["value"]); print({k:(round(v["frac"],3), round(v.get("ms",0),2)) for k,v in j["stage_rooflines"].items()}); print(j["stage_ms_per_step"])
PY
timeout 900 python tools/run_configs.py > gpurun_out/r02/configs.log 2>&1; tail -30 gpurun_out/r02/configs.log
timeout 1500 bash tools/prof_r02.sh > gpurun_out/r02/prof.log 2>&1; tail -40 gpurun_out/prof_r02/summary.log
