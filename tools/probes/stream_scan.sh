#!/bin/bash
# streamed panel tiles (follower workgroup + streamed worker solves + half-tile owners) against the round-3 chain
cd "$(dirname "$0")"
B=./bin/potrf_bench
export POTRF_BENCH_QUICK=1
echo "== default (streamed), with the follower's timeline"; POTRF_BENCH_TRACE=1 POTRF_BENCH_FOLLOWER=1 timeout 120 $B 1024 2048 4096 2>&1 | grep -v "^mask"
for sp in 0 1 2 4 40; do echo "== SLS_POTRF_SPLIT=$sp"; SLS_POTRF_SPLIT=$sp timeout 120 $B 512 1024 2048 3072 4096 2>&1 | grep "dataflow single"; done
echo "== SLS_POTRF_STREAM=0 (round-3 chain)"; SLS_POTRF_STREAM=0 timeout 120 $B 512 1024 2048 3072 4096 8192 2>&1 | grep "dataflow single"
echo "== SLS_POTRF_STREAM=1 at N = 8192"; SLS_POTRF_STREAM=1 timeout 120 $B 8192 2>&1 | grep "dataflow single"
