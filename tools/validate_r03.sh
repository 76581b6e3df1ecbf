#!/bin/bash
# round-3 validation: whole GPU suite + the default bench line (with cpu_baseline + parity)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r03/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/r03/pytest_gpu.log; tail -14 gpurun_out/r03/pytest_gpu.log
cp gpurun_out/c5_full_fit.json gpurun_out/map_optima_report.json gpurun_out/r03/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r03/bench.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "value", j["value"], "frac", j["roofline"]["frac"], "parity", j.get("parity_max_rel"))
print({k:round(v["frac"],3) for k,v in j["stage_rooflines"].items()}); print(j["stage_ms_per_step"]); print(j["cpu_baseline"]["sample"])
PY
