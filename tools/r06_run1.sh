mkdir -p gpurun_out/r06
timeout 600 python bench.py --no-cpu-baseline --no-traffic --steps 2 --warmup 1 > gpurun_out/r06/bench_kmath.json 2>/dev/null
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_kmath.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['roofline']['frac']); print(j['stage_ms_per_step']); print({k:round(v['frac'],3) for k,v in j['stage_rooflines'].items()})
PY
