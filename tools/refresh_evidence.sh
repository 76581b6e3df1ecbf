#!/bin/bash
# refresh round-2 evidence: bench (with cpu baseline), configs, rocprof kernel stats + PMC, default-command kernel stats,
# per-GPU shard shape of the 8-GPU run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python bench.py > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r02/bench_final.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "value", j["value"], "frac", j["roofline"]["frac"], "parity", j.get("parity_max_rel"))
PY
timeout 900 python tools/run_configs.py > gpurun_out/r02/configs.log 2>&1; tail -5 gpurun_out/r02/configs.log
timeout 1500 bash tools/prof_r02.sh > gpurun_out/r02/prof.log 2>&1; tail -4 gpurun_out/prof_r02/summary.log
timeout 600 bash tools/prof_default_cmd.sh 2>&1 | tail -3
for st in 8192 16384 32768; do timeout 300 python bench.py --starts $st --no-cpu-baseline --steps 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$st starts (the per-GPU shard of an 8 / 4 / 2-GPU run):', round(j['ms_per_step'],1), 'ms per step, acq_gemm', round(j['roofline']['frac'],4), 'of peak')"; done > gpurun_out/r02/shard_shapes.log 2>&1; cat gpurun_out/r02/shard_shapes.log
