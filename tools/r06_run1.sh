mkdir -p gpurun_out/r06
cd tools/probes
for v in BASE NORAISE NOSYRK NOSTORE; do
echo "== $v"
(POTRF_BENCH_QUICK=1 POTRF_BENCH_TRACE=1 POTRF_BENCH_FINE=1 timeout 100 ./bin/potrf_bench_$v 2048) 2>&1 | grep -E "dataflow single|fine:|^ +[0-9]+ \|" | head -12
done
