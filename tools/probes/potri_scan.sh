#!/bin/bash
# fused potri (factorisation + inverse in one launch) against the separate launches; team sizes / chunk sizes
cd "$(dirname "$0")"
B=./bin/potrf_bench
export POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1
echo "== defaults"; timeout 120 $B 384 640 1024 2048 3072 4096
for w1 in 80 96 112; do echo "== SLS_POTRI_W1=$w1"; SLS_POTRI_W1=$w1 timeout 120 $B 2048 3072 4096; done
for c in "1 2" "2 1" "2 3"; do set -- $c; echo "== SLS_POTRI_W1=96 CX=$1 CK=$2"; SLS_POTRI_W1=96 SLS_POTRI_CX=$1 SLS_POTRI_CK=$2 timeout 120 $B 2048 4096; done
