mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06/pytest_gpu_c.log 2>&1
tail -8 gpurun_out/r06/pytest_gpu_c.log | cut -c1-300
