#!/bin/bash
# round-2 batch 1: whole GPU suite (with durations), diverging-start diagnostic, default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
nproc > gpurun_out/r02/nproc.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r02/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02/pytest_gpu.log
timeout 300 python tools/diverging_starts.py > gpurun_out/r02/diverging.log 2>&1
timeout 600 python bench.py > gpurun_out/r02/bench1.json 2> gpurun_out/r02/bench1.err
SLS_COMPACT=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02/bench1_nocompact.json 2> gpurun_out/r02/bench1_nocompact.err
tail -5 gpurun_out/r02/pytest_gpu.log
