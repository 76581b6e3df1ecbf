cd tools/probes
export POTRF_BENCH_QUICK=1
for cfg in "X=1" "SLS_POTRF_STREAM=1" "SLS_POTRF_STREAM=1 SLS_POTRF_DNBO=1" "SLS_POTRF_DNBO=1" "SLS_POTRF_DNBO=3" "SLS_POTRF_STREAM=1 SLS_POTRF_SPLIT=1" "SLS_POTRF_STREAM=1 SLS_POTRF_SPLIT=4" "SLS_POTRF_STREAM=1 SLS_POTRF_DNBO=2 SLS_POTRF_DNEAR=2" "SLS_POTRF_DNBO=2 SLS_POTRF_DNEAR=2" "SLS_POTRF_DNBO=4 SLS_POTRF_DNEAR=4"; do
  echo "== $cfg: $(env $cfg timeout 120 ./bin/potrf_bench 8192 2>&1 | grep 'dataflow single' | awk '{print $6, $7}')  | 6144: $(env $cfg timeout 120 ./bin/potrf_bench 6144 2>&1 | grep 'dataflow single' | awk '{print $6, $7}')"
done
