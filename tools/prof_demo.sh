#!/bin/bash
# kernel trace of the two demo scenarios (C1: bayesian_optimization_1d 1 20, C3: sequential_line_search_nd 32 30)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_demo
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for cfg in "c1 bayesian_optimization_1d 1 20 1" "c3 sequential_line_search_nd 32 30"; do
  set -- $cfg; name=$1; shift
  /usr/bin/time -f "$name wall %e s" $R/sequential-line-search_amd/bin/$1 ${@:2} > $OUT/$name.out 2> $OUT/$name.time; tail -1 $OUT/$name.time
  timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/$name -o trace -- $R/sequential-line-search_amd/bin/$1 ${@:2} > $OUT/$name.log 2>&1
  python - <<PY
import sqlite3
c = sqlite3.connect("$OUT/$name/trace_results.db")
tot = 0.0
rows = list(c.execute("select name,total_calls,total_duration,average from top_kernels"))
for r in rows: tot += r[2]
print("$name: %d kernels launched, %.1f ms of kernel time" % (sum(r[1] for r in rows), tot / 1e6))
for r in rows[:10]:
    print("   ", r[0][:60].ljust(60), r[1], "total %.1f ms" % (r[2]/1e6), "avg %.2f us" % (r[3]/1e3))
PY
done
