#!/bin/bash
# round 4: pruned Cholesky schedules + concurrent bordered factorisations: whole GPU suite, probe, C5 numbers
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -q -x --durations=6 > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04/pytest_gpu.log; tail -14 gpurun_out/r04/pytest_gpu.log
cp gpurun_out/c5_full_fit.json gpurun_out/r04/ 2>/dev/null; cat gpurun_out/c5_full_fit.json
cd tools/probes
# (profiles/r04_potrf_chain2.log came from here: SLS_POTRF_DCHAIN2=0/1 before the second chain workgroup was removed)
POTRF_BENCH_QUICK=1 timeout 200 ./bin/potrf_bench 1024 2048 4096 8192 16384 2>&1 | grep -E "dataflow single|one-level" | tee ../../gpurun_out/r04/potrf_sizes.log
POTRF_BENCH_QUICK=1 POTRF_BENCH_BATCH=1 timeout 300 ./bin/potrf_bench 1024 2048 4096 2>&1 | grep -E "batch of|dataflow single" | tee ../../gpurun_out/r04/potrf_batch.log
POTRF_BENCH_QUICK=1 POTRF_BENCH_TRACE=1 POTRF_BENCH_FINE=1 timeout 100 ./bin/potrf_bench 2048 2>&1 | grep -E "dataflow|fine|workers with|chain waited|^ +[0-9]+ \|" | cut -c1-300 | tee ../../gpurun_out/r04/potrf_fine_2048.log
cd ../..
timeout 900 python tools/run_configs.py > gpurun_out/r04/run_configs.log 2>&1; tail -5 gpurun_out/r04/run_configs.log; cp gpurun_out/configs.json gpurun_out/r04/configs.json
