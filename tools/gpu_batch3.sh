#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
for n in 384 1024 2048 4096 8192; do
POTRF_BENCH_TRACE=1 POTRF_BENCH_QUICK=1 timeout 120 tools/probes/bin/potrf_bench $n 2>&1 | grep -v "^mask"
done > gpurun_out/r02/potrf_persist.log 2>&1
for nbo in 1 2 8; do echo "PNBO=$nbo"; SLS_POTRF_PNBO=$nbo POTRF_BENCH_QUICK=1 timeout 120 tools/probes/bin/potrf_bench 8192 2>&1 | grep persistent; done >> gpurun_out/r02/potrf_persist.log 2>&1
cat gpurun_out/r02/potrf_persist.log
