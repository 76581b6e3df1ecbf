#!/bin/bash
# refresh round-2 evidence with gemm_tile_mc: bench (with cpu baseline), configs, rocprof kernel stats + PMC, gate-phase scan
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
for ph in 0 1000 2000 4000; do echo "GATE_PHASE=$ph starts=65536"; SLS_GATE_PHASE=$ph timeout 300 python bench.py --no-cpu-baseline --steps 1 --n-local 8 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['frac'])"; done > gpurun_out/r02/gate_phase_65536.log 2>&1; cat gpurun_out/r02/gate_phase_65536.log
for ph in 0 2000 4000; do echo "GATE_PHASE=$ph starts=8192"; SLS_GATE_PHASE=$ph timeout 300 python bench.py --starts 8192 --no-cpu-baseline --steps 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['frac'])"; done > gpurun_out/r02/gate_phase_8192.log 2>&1; cat gpurun_out/r02/gate_phase_8192.log
