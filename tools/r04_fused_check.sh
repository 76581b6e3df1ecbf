#!/bin/bash
# round 4 (second session): whole GPU suite on the fused factor + inverse launch and the streamed chain, configs, probe numbers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
date +%T
( cd tools/probes && POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 timeout 200 ./bin/potrf_bench 384 640 1024 1536 2048 3072 4096 2>&1 | grep -v "^mask\|   U " ) > gpurun_out/r04b_potri_probe.log 2>&1; grep "fused single\|trtri\|dataflow single" gpurun_out/r04b_potri_probe.log | cut -c1-100
date +%T
timeout 1500 python -u -m pytest tests -m gpu -q --durations=8 --timeout 400 > gpurun_out/r04b_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04b_pytest_gpu.log; tail -25 gpurun_out/r04b_pytest_gpu.log | cut -c1-250
date +%T
timeout 200 python -u tools/run_configs.py > gpurun_out/r04_configs_fused.json 2> gpurun_out/r04_configs_fused.err; tail -3 gpurun_out/r04_configs_fused.err
date +%T
