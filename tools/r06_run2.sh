mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests/test_gpu_bench.py tests/test_gpu_early_stop.py tests/test_gpu_map_device.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r06/pytest_gpu_b.log 2>&1
tail -30 gpurun_out/r06/pytest_gpu_b.log | cut -c1-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/early_stop_single_start.json'))
print({k:v for k,v in d.items() if k!='rows'})
for r in sorted(d['rows'], key=lambda r:-r['loss_rel'])[:8]: print(r)
PY
