#!/bin/bash
# new GEMM tile (gemm_tile_mc) in the library: whole GPU suite, bench, potrf probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -q -x --durations=4 > gpurun_out/r02/pytest_mc.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02/pytest_mc.log; tail -8 gpurun_out/r02/pytest_mc.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_mc.json 2> gpurun_out/r02/bench_mc.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r02/bench_mc.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "value", j["value"], "frac", j["roofline"]["frac"])
print({k:round(v["frac"],3) for k,v in j["stage_rooflines"].items()}); print(j["stage_ms_per_step"])
PY
timeout 300 python bench.py --starts 8192 --no-cpu-baseline --steps 2 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8192 starts', j['ms_per_step'], j['roofline']['frac'])"
cd tools/probes && for n in 2048 4096 8192; do timeout 120 ./bin/potrf_bench $n 2>&1 | tail -3; done
