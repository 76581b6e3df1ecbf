#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export POTRF_BENCH_HYBRID=1 POTRF_BENCH_QUICK=1
for h in 2 4 8; do echo "HNBO=$h"; SLS_POTRF_MODE=2 SLS_POTRF_HNBO=$h timeout 120 tools/probes/bin/potrf_bench 2048 4096 8192 16384 2>&1 | grep -v "mask\|single"; done > gpurun_out/r02/potrf_hybrid.log 2>&1
cat gpurun_out/r02/potrf_hybrid.log
