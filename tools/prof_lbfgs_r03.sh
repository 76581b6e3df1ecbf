#!/bin/bash
# per-dispatch durations of the L-BFGS step kernels (register form / memory form) against the number of live starts
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03
cd /tmp
for f in 1 0; do
  OUT=/tmp/prof_lb$f; rm -rf $OUT; mkdir -p $OUT
  SLS_LBFGS_REG=$f rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/out.log 2> $OUT/err.log
  t=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python3 - "$t" $f <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lbfgs_step" in r["Kernel_Name"]]
print("REG=%s: %d dispatches" % (sys.argv[2], len(rows)))
for r in rows[:50]:
    g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    print("  live<=%6d  %8.1f us" % (g // 64 * 16, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
done
