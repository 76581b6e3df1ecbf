#!/bin/bash
# fused potri (factorisation + inverse in one launch) against the separate launches; team sizes / chunk sizes / split-off last term
cd "$(dirname "$0")"
B=./bin/potrf_bench
export POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1
f() { grep "fused single\|trtri\|K^-1" | grep -v "same bits"; }
echo "== defaults"; timeout 120 $B 384 640 1024 2048 3072 4096 | f
for w1 in 64 80 112 128; do echo "== SLS_POTRI_W1=$w1"; SLS_POTRI_W1=$w1 timeout 120 $B 2048 4096 | f; done
for c in 1 2 4; do echo "== SLS_POTRI_CX=$c SLS_POTRI_CK=$c"; SLS_POTRI_CX=$c SLS_POTRI_CK=$c timeout 120 $B 2048 4096 | f; done
for p in 0 1; do echo "== SLS_POTRI_PLAST=$p"; SLS_POTRI_PLAST=$p timeout 120 $B 1024 2048 3072 4096 | f; done
for sp in 1 2 40; do echo "== SLS_POTRF_SPLIT=$sp"; SLS_POTRF_SPLIT=$sp timeout 120 $B 2048 3072 4096 | f; done
