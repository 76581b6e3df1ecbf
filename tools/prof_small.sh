#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_small
rm -rf $OUT; mkdir -p $OUT
cd /tmp
SLS_TIME_FLAGS=0 timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/time_small.py > $OUT/trace.log 2>&1
python - <<PY
import sqlite3
c = sqlite3.connect("$OUT/trace/trace_results.db")
for r in c.execute("select name,total_calls,total_duration,average from top_kernels limit 12"):
    print(r[0][:60].ljust(60), r[1], "total %.1f ms" % (r[2]/1e6), "avg %.2f us" % (r[3]/1e3))
PY
