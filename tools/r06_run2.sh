mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06/pytest_gpu_a.log 2>&1
tail -5 gpurun_out/r06/pytest_gpu_a.log
timeout 600 python bench.py > gpurun_out/r06/bench_a.json 2> gpurun_out/r06/bench_a.err
tail -c 3000 gpurun_out/r06/bench_a.json
