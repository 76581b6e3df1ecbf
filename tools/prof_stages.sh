#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_stages
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/time_stages.py > $OUT/trace.log 2>&1
python - <<PY
import sqlite3
c = sqlite3.connect("$OUT/trace/trace_results.db")
for r in c.execute("select name,total_calls,total_duration,average from top_kernels limit 14"):
    print(r[0][:70].ljust(70), r[1], round(r[2]/1e3,1), "us total", round(r[3]/1e3,2), "us avg")
PY
