#!/bin/bash
# round 4 (second session): whole GPU suite on the fused factor + inverse launch, exit-order probe, configs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
date +%T
timeout 1500 python -u -m pytest tests -m gpu -q --durations=8 --timeout 400 > gpurun_out/r04b_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04b_pytest_gpu.log; tail -25 gpurun_out/r04b_pytest_gpu.log | cut -c1-250
date +%T
# handles alive at interpreter exit, context collected in whatever order the interpreter picks: must exit by itself
timeout 60 python -u - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from util import sls, synth_problem
from oracle import oracle_py as oracle
m = sls(); ctx = m.Context(0)
X, y, theta, b = synth_problem(oracle, 8, 600)
g = m.GP(ctx, X, y, theta, b, 1); h = m.Nll(ctx, X, 1)
print("value", h.gp_objective(y, np.concatenate([[0.5, 0.01], np.full(8, 0.5)]))[0], "exiting without close()")
PY
echo "exit-order probe: exit code $? (124 = hung)"
date +%T
timeout 200 python -u tools/run_configs.py > gpurun_out/r04_configs_fused.json 2> gpurun_out/r04_configs_fused.err; tail -3 gpurun_out/r04_configs_fused.err
date +%T
