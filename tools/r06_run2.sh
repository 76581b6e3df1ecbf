mkdir -p gpurun_out/r06
SLS_TEST_EXTRA_SEEDS=150 timeout 1800 python -m pytest tests/test_gpu_stress.py -q -m gpu > gpurun_out/r06/stress_sweep.log 2>&1
tail -3 gpurun_out/r06/stress_sweep.log
cd tools/probes
(POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 POTRF_BENCH_STRESS=1500 timeout 900 ./bin/potrf_bench 1536 1920 2048 2304 2560) 2>&1 | grep -E "stress|potri fused" > ../../gpurun_out/r06/potri_stress_default.log
cat ../../gpurun_out/r06/potri_stress_default.log
