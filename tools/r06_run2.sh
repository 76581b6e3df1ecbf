mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06/pytest_gpu_d.log 2>&1
tail -8 gpurun_out/r06/pytest_gpu_d.log | cut -c1-300
