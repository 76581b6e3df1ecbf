mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06/pytest_gpu_f.log 2>&1
grep -E "passed|failed" gpurun_out/r06/pytest_gpu_f.log | tail -2; grep "^FAILED" gpurun_out/r06/pytest_gpu_f.log | head
python tools/prof_c5.py 2>&1 | tail -2
