#!/bin/bash
# fused potri (factorisation + inverse in one launch) against the separate launches; team sizes / chunk sizes / K^-1 rows shared
# with the factorisation's workers
cd "$(dirname "$0")"
B=./bin/potrf_bench
export POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1
f() { grep "fused single\|trtri\|K^-1" | grep -v "same bits"; }
echo "== defaults"; timeout 120 $B 384 640 1024 2048 3072 4096 | f
for ks in 12 14; do echo "== N=2048 KSPLIT=$ks"; SLS_POTRI_KSPLIT=$ks timeout 120 $B 2048 | f; done
for ks in 16 20 24; do echo "== N=3072/4096 KSPLIT=$ks (3072: nb=24)"; SLS_POTRI_KSPLIT=$ks timeout 120 $B 3072 4096 | f; done
for w1 in 64 128; do echo "== KSPLIT=20 W1=$w1"; SLS_POTRI_KSPLIT=20 SLS_POTRI_W1=$w1 timeout 120 $B 4096 | f; done
echo "== KSPLIT=20 CK=1"; SLS_POTRI_KSPLIT=20 SLS_POTRI_CK=1 timeout 120 $B 4096 | f
