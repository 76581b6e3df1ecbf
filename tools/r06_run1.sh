mkdir -p gpurun_out/r06
cd tools/probes
for P in 0 1 2 3 4; do
echo "== prio $P"
(SLS_POTRF_PRIO=$P POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1 timeout 300 ./bin/potrf_bench 2048 3072 4096 8192) 2>&1 | grep -E "dataflow single|potri fused" | cut -c1-100
done > ../../gpurun_out/r06/potrf_prio_scan.log 2>&1
cat ../../gpurun_out/r06/potrf_prio_scan.log
