#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (python bench.py: 2 timed steps + 1 warm-up, 65 536 starts, cap 50;
# --no-cpu-baseline only skips the host-side oracle timing): the per-kernel averages the bench line's roofline.avg_launch_ms
# must agree with.  Writes gpurun_out/r03/r03_kernel_stats_default_cmd.csv
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/prof_default; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/r03
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $R/bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" $R/gpurun_out/r03/r03_kernel_stats_default_cmd.csv $OUT/bench.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    o.write("kernel,calls,total_us,avg_us,percent\n")
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "").replace("slsk::", "")
        o.write(f'"{name}",{r["Calls"]},{float(r["TotalDurationNs"])/1e3:.1f},{float(r["AverageNs"])/1e3:.2f},{r["Percentage"]}\n')
j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("bench line: ms/step", j["ms_per_step"], "avg_launch_ms", j["roofline"]["avg_launch_ms"], "launches", j["roofline"]["launches"])
for r in rows[:4]: print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6, "ms avg")
PY
