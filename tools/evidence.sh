#!/bin/bash
# Evidence of one round for profiles/ (run on the GPU box: gpurun -- 'bash tools/evidence.sh r05'):
#   <tag>_bench_line_1gpu.json            the DEFAULT bench.py line (cpu_baseline, parity and the run's own traffic measurement included)
#   <tag>_kernel_stats_default_cmd.csv    rocprofv3 --kernel-trace --stats of the same command (without the CPU leg / the PMC children)
#   <tag>_pmc_acq_gemm.json               MFMA-pipe counters of the full-shape acq_gemm launch (own pass, --kernel-trace only beside --pmc)
#   <tag>_kernel_stats_c5.csv             kernel stats of the C5 MAP evaluation (objective + gradient; batched value-only evaluations)
#   <tag>_configs.json                    tools/run_configs.py: C1, C2, C3 (both hyper-parameter variants), C5 with their CPU legs
# Nothing here reads /root/reference.
TAG=${1:-r05}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/${TAG}_bench_line_1gpu.json 2> $O/bench.err
python - "$O/${TAG}_bench_line_1gpu.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("ms/step", j["ms_per_step"], "value", j["value"], "frac", r["frac"], "traffic GB", (r["traffic"] or 0) / 1e9, "parity", j.get("parity_max_rel"),
      "fallbacks", j.get("potrf_fallbacks"))
print({k: round(v["frac"], 3) for k, v in j["stage_rooflines"].items()})
PY
P=/tmp/prof_default; rm -rf $P; mkdir -p $P
cd /tmp
rocprofv3 --kernel-trace --stats -d $P -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-traffic > $P/bench.json 2> $P/err.log
python3 - "$(find $P -name '*kernel_stats.csv' | head -1)" $O/${TAG}_kernel_stats_default_cmd.csv $P/bench.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    o.write("kernel,calls,total_us,avg_us,percent\n")
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "").replace("slsk::", "")
        o.write(f'"{name}",{r["Calls"]},{float(r["TotalDurationNs"]) / 1e3:.1f},{float(r["AverageNs"]) / 1e3:.2f},{r["Percentage"]}\n')
j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("bench line under rocprof: ms/step", j["ms_per_step"], "avg_launch_ms", j["roofline"]["avg_launch_ms"], "launches", j["roofline"]["launches"])
for r in rows[:4]:
    print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, "ms avg")
PY
P=/tmp/prof_pmc; rm -rf $P; mkdir -p $P
SLS_COMPACT=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 -d $P -o pmc --output-format csv -- \
    python $R/bench.py --traffic-child > $P/log.txt 2>&1
python3 - "$(find $P -name '*counter_collection.csv' | head -1)" $O/${TAG}_pmc_acq_gemm.json <<'PY'
import collections, csv, json, sys
per = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    if "acq_gemm_kernel" in r["Kernel_Name"]:
        d = per[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
m = lambda k: sum(d[k] for d in per.values()) / len(per)
gui = m("GRBM_GUI_ACTIVE") / 8                      # summed over the 8 XCDs
flops = m("SQ_INSTS_VALU_MFMA_MOPS_F64") * 512
out = {"kernel": "acq_gemm_kernel", "launches_profiled": len(per), "avg_duration_ms": m("ns") / 1e6, "mfma_flops_counted": flops,
       "candidates_per_launch": int(round(flops / (2.0 * 8192 * 8192))), "effective_clock_GHz": gui / m("ns"),
       "mfma_busy_fraction": m("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / gui,     # summed over 256 CU x 4 SIMD
       "mfma_busy_cycles_per_instruction": m("SQ_VALU_MFMA_BUSY_CYCLES") / (flops / 2048),
       "command": "SLS_COMPACT=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 "
                  "-- python bench.py --traffic-child (tools/evidence.sh); memory-side bytes: roofline.traffic of the bench line itself"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print({k: out[k] for k in ("avg_duration_ms", "mfma_busy_fraction", "effective_clock_GHz", "candidates_per_launch")})
PY
P=/tmp/prof_c5; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats -d $P -o t --output-format csv -- python $R/tools/prof_c5.py > $P/log.txt 2>&1
python3 - "$(find $P -name '*kernel_stats.csv' | head -1)" $O/${TAG}_kernel_stats_c5.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as o:
    o.write("kernel,calls,total_us,avg_us,percent\n")
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "").replace("slsk::", "")
        o.write(f'"{name}",{r["Calls"]},{float(r["TotalDurationNs"]) / 1e3:.1f},{float(r["AverageNs"]) / 1e3:.2f},{r["Percentage"]}\n')
for r in rows[:8]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg")
PY
cd $R
timeout 900 python tools/run_configs.py > $O/run_configs.log 2>&1 && cp gpurun_out/configs.json $O/${TAG}_configs.json
tail -3 $O/run_configs.log | cut -c1-300
