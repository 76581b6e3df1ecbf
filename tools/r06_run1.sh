mkdir -p gpurun_out/r06
cd tools/probes
(SLS_POTRF_FUSE_SYRK=1 SLS_POTRF_LU_W=0 POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 POTRF_BENCH_STRESS=1000 timeout 900 ./bin/potrf_bench 1536 2048 2560 3072 4096) > ../../gpurun_out/r06/potrf_chain3_drain.log 2>&1
grep -E "dataflow single|potri fused|stress" ../../gpurun_out/r06/potrf_chain3_drain.log | cut -c1-120
(SLS_POTRF_FUSE_SYRK=1 SLS_POTRF_LU_W=6 POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 POTRF_BENCH_STRESS=1000 timeout 900 ./bin/potrf_bench 2560 3072) 2>&1 | grep -E "dataflow single|potri fused|stress" | cut -c1-120
