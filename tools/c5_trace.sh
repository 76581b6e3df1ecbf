#!/bin/bash
# Kernel timeline of the C5 evaluation (N = 4096, D = 128): rocprofv3 kernel trace -> per-kernel start/end of the LAST evaluations,
# with the idle gap before each kernel (tools/c5_trace_gaps.py).  Output: gpurun_out/c5_trace_gaps.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/c5trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/c5trace -- python -u $R/tools/c5_eval_ab.py > /tmp/c5trace.log 2>&1
tail -2 /tmp/c5trace.log
f=$(find /tmp/c5trace -name '*kernel_trace.csv' | head -1)
python $R/tools/c5_trace_gaps.py "$f" > $R/gpurun_out/c5_trace_gaps.txt
cat $R/gpurun_out/c5_trace_gaps.txt
