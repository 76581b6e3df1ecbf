#!/bin/bash
# second parameter scan of the dataflow Cholesky (TRSM chain): chunk length / near rule / owner map at N = 8192 and 16384
run() { echo "== $*"; env "$@" POTRF_BENCH_QUICK=1 POTRF_BENCH_TRACE=1 timeout 100 ./bin/potrf_bench $SIZES | grep -E "dataflow single|workers with|chain waited" | cut -c1-330; }
SIZES="8192"
run SLS_POTRF_DNBO=2
run SLS_POTRF_DNBO=3
run SLS_POTRF_DNBO=4
run SLS_POTRF_DNBO=4 SLS_POTRF_DNEAR=2
run SLS_POTRF_DNBO=4 SLS_POTRF_DNEAR=4
run SLS_POTRF_DNBO=8 SLS_POTRF_DNEAR=4
run SLS_POTRF_DNBO=2 SLS_POTRF_DACQ=0
SIZES="16384"
run SLS_POTRF_DNBO=2
run SLS_POTRF_DNBO=4
run SLS_POTRF_DNBO=4 SLS_POTRF_DNEAR=2
run SLS_POTRF_DNBO=8 SLS_POTRF_DNEAR=3
SIZES="4096"
run SLS_POTRF_DNBO=1
run SLS_POTRF_DNBO=2
