#!/bin/bash
# per-dispatch durations of one C5 MAP evaluation (N = 4096, D = 128): which launches of the fit pipeline take how long
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/prof_c5t; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/r03
cd /tmp
rocprofv3 --kernel-trace -d $OUT -o t --output-format csv -- python $R/tools/prof_c5.py > $OUT/out.log 2> $OUT/err.log
t=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last evaluation: from the last prep_kernel on
last = max(i for i, r in enumerate(rows) if "prep_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slsk::", "")[:44]
    print("%9.1f us  +%7.1f us  grid %7s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "?"), name))
PY
