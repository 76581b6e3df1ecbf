#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_host_cpp.py tests/test_gpu_bench.py -m gpu -q -x --durations=10 > gpurun_out/r02/pytest_b2.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02/pytest_b2.log
timeout 600 tools/probes/bin/potrf_bench 2048 4096 8192 > gpurun_out/r02/potrf_bench.log 2>&1
timeout 600 bash tools/kat_1d.sh > gpurun_out/r02/kat_1d.log 2>&1
tail -3 gpurun_out/r02/pytest_b2.log; grep KAT gpurun_out/r02/kat_1d.log; cat gpurun_out/r02/potrf_bench.log
