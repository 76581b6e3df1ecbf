mkdir -p gpurun_out/r06
cd tools/probes
(POTRF_BENCH_QUICK=1 POTRF_BENCH_TRACE=1 POTRF_BENCH_POTRI=1 POTRF_BENCH_STRESS=1500 timeout 1200 ./bin/potrf_bench 2048 2816 3072 3584 4096) 2>&1 | grep -E "potri fused|fused trace|stress|max.fused" | cut -c1-200 > ../../gpurun_out/r06/potri_pool.log
cat ../../gpurun_out/r06/potri_pool.log
cd ../..
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py tests/test_gpu_budgets.py -x -q -m gpu 2>&1 | tail -3
