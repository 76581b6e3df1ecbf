#!/bin/bash
# Quick look at config C3 (sequential_line_search_nd 32 30) on the GPU box: steady ms per SubmitFeedbackData in both hyper-parameter
# variants, the host split and the section traces of the two one-workgroup kernels.  Optional: directories holding alternative builds
# of libsls_hip.so (LD_LIBRARY_PATH precedes the binaries' RUNPATH) are measured one after the other.
#   gpurun -- 'bash tools/c3_quick.sh [variant_dir ...]'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
B=./sequential-line-search_amd/bin/sequential_line_search_nd
one() {
  for m in 1 0; do
    for rep in 1 2 3; do $B 32 30 1 $m | awk -v m=$m 'NR>1{s+=$NF;n++}END{printf "MAP=%d steady mean %.3f ms over %d submits\n", m, s/n, n}'; done
  done
  SLS_HOST_TIMING=1 $B 32 30 1 1 2>&1 | grep "SubmitFeedbackData" | tail -3
  SLS_HOST_TIMING=1 $B 32 30 1 0 2>&1 | grep "SubmitFeedbackData\|FindNextPointDirect" | tail -6
  SLS_MAP_TRACE=1 $B 32 30 1 1 2>&1 | grep -A1 "map_opt trace" | grep -v "^--" | tail -4
  SLS_MAP_TRACE=1 $B 32 30 1 0 2>&1 | grep -A1 "map_opt trace" | grep -v "^--" | tail -2
  SLS_WAVE_TRACE=1 $B 32 30 1 1 2>&1 | grep "wave trace" | tail -2
}
echo "=== shipped build"; one
for v in "$@"; do
  echo "=== variant $v"
  LD_LIBRARY_PATH=$R/$v:$LD_LIBRARY_PATH one
done
