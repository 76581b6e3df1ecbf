#!/bin/bash
# Probe logs of a round for profiles/ (after `make -C tools/probes`): factorisation timings, the diagonal tile's cycle stamps, the
# latencies that price the one-workgroup kernels, section traces of config C3 in both hyper-parameter variants, the counters of
# map_opt_kernel, the GPU suite.   gpurun -- 'bash tools/evidence_probes.sh r05'
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R/tools/probes
(export POTRF_BENCH_QUICK=1 POTRF_BENCH_POTRI=1; timeout 300 ./bin/potrf_bench 384 1024 2048 3072 4096 2>&1 | grep -v "^mask"; POTRF_BENCH_QUICK=1 timeout 200 ./bin/potrf_bench 8192 16384 2>&1 | grep -v "^mask") > $O/${TAG}_potrf_probe.log 2>&1
timeout 120 ./bin/diag_timing > $O/${TAG}_diag16_timing.log 2>&1
timeout 60 ./bin/lat_probe > $O/${TAG}_lat_probe.log 2>&1
cd $R
B=./sequential-line-search_amd/bin/sequential_line_search_nd
{
  for m in 1 0; do
    echo "== sequential_line_search_nd 32 30 1 $m (use_MAP_hyperparams = $m): ms per SubmitFeedbackData"
    for rep in 1 2 3; do $B 32 30 1 $m | awk 'NR>1{s+=$NF;n++}END{printf "steady mean %.3f ms over %d submits\n", s/n, n}'; done
    echo "== host split (SLS_HOST_TIMING=1), last six submits"
    SLS_HOST_TIMING=1 $B 32 30 1 $m 2>&1 | grep "SubmitFeedbackData\|FindNextPointDirect" | tail -12
  done
  echo "== map_opt_kernel sections (SLS_MAP_TRACE=1), use_MAP_hyperparams = 1, every fourth submit"
  SLS_MAP_TRACE=1 $B 32 30 1 1 2>&1 | grep -A1 "map_opt trace" | grep -v "^--" | awk 'NR%8<2'
  echo "== maximize_wave_kernel sections of the local phase (SLS_WAVE_TRACE=1), last four"
  SLS_WAVE_TRACE=1 $B 32 30 1 1 2>&1 | grep "wave trace" | tail -4
  echo "== concurrent PredictMu"
  ./sequential-line-search_amd/bin/test_host | grep -i "concurrent\|HOST TESTS"
} > $O/${TAG}_c3_trace.log 2>&1
bash tools/pmc_map_opt.sh > $O/${TAG}_pmc_map_opt.log 2>&1
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 > $O/${TAG}_pytest_gpu.log
cp gpurun_out/test_evidence.json $O/${TAG}_test_evidence.json 2>/dev/null
tail -3 $O/${TAG}_pytest_gpu.log; grep "steady mean" $O/${TAG}_c3_trace.log
