#!/bin/bash
# rocprofv3 --kernel-trace --stats of the C5 MAP objective + gradient (N = 4096, D = 128, Matern-5/2; tools/prof_c5.py)
# -> gpurun_out/r03/r03_kernel_stats_c5.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/prof_c5; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/r03
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $R/tools/prof_c5.py > $OUT/out.log 2> $OUT/err.log
cat $OUT/out.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" $R/gpurun_out/r03/r03_kernel_stats_c5.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    o.write("kernel,calls,total_us,avg_us,percent\n")
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "").replace("slsk::", "")
        o.write(f'"{name}",{r["Calls"]},{float(r["TotalDurationNs"])/1e3:.1f},{float(r["AverageNs"])/1e3:.2f},{r["Percentage"]}\n')
print(open(sys.argv[2]).read())
PY
