#!/bin/bash
# round-3 evidence with the final code: GPU suite, default bench line (cpu_baseline + parity), configs, rocprof kernel stats (default
# command, C5) + PMC passes, Cholesky probe (sizes, trace), per-GPU shard shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r03/pytest_gpu_final.log 2>&1
echo "pytest exit $?" >> gpurun_out/r03/pytest_gpu_final.log; grep -E "passed|failed" gpurun_out/r03/pytest_gpu_final.log | tail -2
cp gpurun_out/c5_full_fit.json gpurun_out/map_optima_report.json gpurun_out/test_evidence.json gpurun_out/r03/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r03/bench_final.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "value", j["value"], "frac", j["roofline"]["frac"], "parity", j.get("parity_max_rel"))
print({k:round(v["frac"],3) for k,v in j["stage_rooflines"].items()}); print(j["stage_ms_per_step"]); print(j["cpu_baseline"]["sample"])
PY
timeout 600 python tools/run_configs.py > gpurun_out/r03/configs.log 2>&1; cp gpurun_out/configs.json gpurun_out/r03/configs.json; tail -30 gpurun_out/r03/configs.log
timeout 900 bash tools/prof_default_cmd_r03.sh 2>&1 | tail -4
timeout 600 bash tools/prof_c5_r03.sh 2>&1 | head -14
timeout 1400 bash tools/prof_r03.sh > gpurun_out/r03/prof.log 2>&1; tail -3 gpurun_out/prof_r03/summary.log
(cd tools/probes && POTRF_BENCH_TRACE=1 POTRF_BENCH_QUICK=1 timeout 200 ./bin/potrf_bench 1024 2048 4096 8192 16384 > ../../gpurun_out/r03/potrf_final.log 2>&1; grep -E "dataflow single|persistent single|one-level" ../../gpurun_out/r03/potrf_final.log)
timeout 1200 python tools/shard_shapes.py > gpurun_out/r03/shard_shapes.log 2>&1; cp gpurun_out/shard_shapes.json gpurun_out/r03/; grep -E "starts_per_gpu|ms_per_step|single_gpu_step" gpurun_out/r03/shard_shapes.log
