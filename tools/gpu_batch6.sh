#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -q -x --durations=4 > gpurun_out/r02/pytest_b6.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02/pytest_b6.log; tail -8 gpurun_out/r02/pytest_b6.log
timeout 900 python tools/run_configs.py > gpurun_out/r02/configs.log 2>&1; grep -A12 "C2_" gpurun_out/r02/configs.log | head -20;  grep -A3 "C5_\|C1_" gpurun_out/r02/configs.log | head
