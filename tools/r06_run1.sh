cd tools/probes
(POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 timeout 300 ./bin/potrf_bench 2816 3072 3584 4096) 2>&1 | grep -E "potri fused" | cut -c1-100
(SLS_POTRI_POOL=1 POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 timeout 300 ./bin/potrf_bench 2048 2560) 2>&1 | grep -E "potri fused" | cut -c1-100
