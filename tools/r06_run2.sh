mkdir -p gpurun_out/r06
cd tools/probes
(POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 POTRF_BENCH_STRESS=1000 timeout 900 ./bin/potrf_bench 1536 2048 2560) 2>&1 | grep -E "dataflow single|stress|potri fused" | cut -c1-110
cd ../..
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06/pytest_gpu_e.log 2>&1
tail -3 gpurun_out/r06/pytest_gpu_e.log | cut -c1-200
