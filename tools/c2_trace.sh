#!/bin/bash
# Kernel timeline of the C2 fit (N = 2048, D = 16): rocprofv3 kernel trace -> idle gap before each kernel of one fit
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/c2trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/c2trace -- python -u $R/tools/c2_fit_ab.py > /tmp/c2trace.log 2>&1
tail -1 /tmp/c2trace.log
f=$(find /tmp/c2trace -name '*kernel_trace.csv' | head -1)
python $R/tools/c5_trace_gaps.py "$f" 8 > $R/gpurun_out/c2_trace_gaps.txt
cat $R/gpurun_out/c2_trace_gaps.txt
