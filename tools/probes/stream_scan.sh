#!/bin/bash
# streamed panel tiles (follower workgroup + streamed worker solves + half-tile owners of the sub-diagonal tiles) against the round-3 chain
cd "$(dirname "$0")"
B=./bin/potrf_bench
export POTRF_BENCH_QUICK=1
echo "== SLS_POTRF_STREAM=1 (split, rows 1)"; SLS_POTRF_STREAM=1 POTRF_BENCH_TRACE=1 timeout 120 $B 1024 2048 4096 2>&1 | grep -v "^mask"
for sp in 0 1; do for r in 1 8; do echo "== SLS_POTRF_STREAM=1 SPLIT=$sp ROWS=$r"; SLS_POTRF_STREAM=1 SLS_POTRF_SPLIT=$sp SLS_POTRF_STREAM_ROWS=$r timeout 120 $B 512 1024 2048 3072 4096 8192 2>&1 | grep "dataflow single"; done; done
echo "== SLS_POTRF_STREAM=0"; SLS_POTRF_STREAM=0 timeout 120 $B 512 1024 2048 3072 4096 8192 2>&1 | grep "dataflow single"
echo "== potri with stream"; SLS_POTRF_STREAM=1 SLS_POTRF_STREAM_ROWS=8 POTRF_BENCH_POTRI=1 timeout 120 $B 2048 4096 2>&1 | grep -v "^mask\|   U "
