#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02/pytest_gpu2.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02/pytest_gpu2.log
tail -14 gpurun_out/r02/pytest_gpu2.log
timeout 600 python tools/diverging_starts.py > gpurun_out/r02/diverging.log 2>&1; cat gpurun_out/r02/diverging.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02/bench2.json 2> gpurun_out/r02/bench2.err; tail -c 1500 gpurun_out/r02/bench2.json
