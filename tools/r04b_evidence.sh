#!/bin/bash
# round-4 (second session) evidence for profiles/: the default bench line (with cpu_baseline + parity), rocprofv3 --kernel-trace --stats of the
# same command, and the kernel stats of the C5 MAP evaluation (objective + gradient, and the batched value-only evaluations)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04b
cd $R
timeout 900 python bench.py > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r04b/bench.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "value", j["value"], "frac", j["roofline"]["frac"], "parity", j.get("parity_max_rel"), "fallbacks", j.get("potrf_fallbacks"))
print({k:round(v["frac"],3) for k,v in j["stage_rooflines"].items()}); print(j["stage_ms_per_step"]); print(j["cpu_baseline"]["sample"])
PY
OUT=/tmp/prof_default; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python $R/bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" $R/gpurun_out/r04b/r04b_kernel_stats_default_cmd.csv $OUT/bench.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    o.write("kernel,calls,total_us,avg_us,percent\n")
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "").replace("slsk::", "")
        o.write(f'"{name}",{r["Calls"]},{float(r["TotalDurationNs"])/1e3:.1f},{float(r["AverageNs"])/1e3:.2f},{r["Percentage"]}\n')
j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("bench line under rocprof: ms/step", j["ms_per_step"], "avg_launch_ms", j["roofline"]["avg_launch_ms"], "launches", j["roofline"]["launches"])
for r in rows[:5]: print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6, "ms avg")
PY
OUT=/tmp/prof_c5; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/c5_both.py <<'PY'
import importlib, os, sys
import numpy as np
R = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, R)
sls = importlib.import_module("sequential-line-search_amd")
N, D = 4096, 128
rng = np.random.default_rng(1234)
X = np.asfortranarray(rng.uniform(0, 1, (D, N)))
y = np.exp(-np.sum((X - 0.4) ** 2, axis=0)) + 0.01 * rng.standard_normal(N)
ctx = sls.Context(0); nll = sls.Nll(ctx, X, sls.KERNEL_MATERN52)
x = np.concatenate([[0.5, 0.005], np.full(D, 0.5 * np.sqrt(D / 8.0))])
for i in range(6): nll.gp_objective(y, x * (1.0 + 0.01 * i))
xs = np.tile(x, (10, 1)); xs[:, 2] *= 1 + 1e-3 * np.arange(10)
for i in range(3): nll.gp_objective_batch(y, xs * (1.0 + 0.01 * i))
PY
rocprofv3 --kernel-trace --stats -d $OUT -o t --output-format csv -- python /tmp/c5_both.py > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" $R/gpurun_out/r04b/r04b_kernel_stats_c5.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    o.write("kernel,calls,total_us,avg_us,percent\n")
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "").replace("slsk::", "")
        o.write(f'"{name}",{r["Calls"]},{float(r["TotalDurationNs"])/1e3:.1f},{float(r["AverageNs"])/1e3:.2f},{r["Percentage"]}\n')
for r in rows[:10]: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg")
PY
