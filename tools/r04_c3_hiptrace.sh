#!/bin/bash
# HIP API + kernel statistics of the C3 demo (where does a SubmitFeedbackData's host time go?)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04/c3_hiptrace
export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --stats -d gpurun_out/r04/c3_hiptrace -o c3 --output-format csv -- sequential-line-search_amd/bin/sequential_line_search_nd 32 30 1 > gpurun_out/r04/c3_hiptrace/run.log 2>&1
find gpurun_out/r04/c3_hiptrace -name "*stats*" | head
for f in $(find gpurun_out/r04/c3_hiptrace -name "*hip_api_stats.csv") $(find gpurun_out/r04/c3_hiptrace -name "*kernel_stats.csv"); do echo "== $f"; head -25 $f; done
find gpurun_out/r04/c3_hiptrace -name "*_trace.csv" -size +1M -delete
