#!/bin/bash
# round 4: wave-kernel / MAP-kernel latency work -- parity subset + C3 / C1 timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_map_device.py tests/test_gpu_host_cpp.py -q -x > gpurun_out/r04/parity_subset.log 2>&1; echo "exit $?" >> gpurun_out/r04/parity_subset.log; tail -6 gpurun_out/r04/parity_subset.log
python tools/time_map_fit.py 2>&1 | grep "one launch" | tee gpurun_out/r04/time_map_fit.log
B=sequential-line-search_amd/bin
for i in 1 2; do SLS_HOST_TIMING=1 $B/sequential_line_search_nd 32 30 1 > gpurun_out/r04/c3_run$i.log 2>&1; done
SLS_MAP_TRACE=1 SLS_WAVE_TRACE=1 SLS_HOST_TIMING=1 $B/sequential_line_search_nd 32 30 1 > gpurun_out/r04/c3_trace.log 2>&1; grep "wave trace\|map_opt trace" gpurun_out/r04/c3_trace.log | tail -6
python - <<'PY'
import re,statistics
for f in ("c3_run1","c3_run2"):
    t=open(f"gpurun_out/r04/{f}.log").read()
    ms=[float(v) for v in re.findall(r" ms ([-\d.e]+)",t)]
    fit=[float(v) for v in re.findall(r"MAP fit ([\d.]+) ms",t)]; nx=[float(v) for v in re.findall(r"next point ([\d.]+) ms",t)]
    print(f,"mean w/o first",statistics.mean(ms[1:]),"median",statistics.median(ms),"max",max(ms[1:]),"map fit mean",statistics.mean(fit[1:]),"next point mean",statistics.mean(nx[1:]))
PY
for i in 1 2 3; do python - <<'PY'
import subprocess,time
t=time.perf_counter(); p=subprocess.run(["sequential-line-search_amd/bin/bayesian_optimization_1d","1","20","1"],capture_output=True,text=True); print("C1 wall %.3f s"%(time.perf_counter()-t), p.stdout.strip().splitlines()[-1])
PY
done
echo "--- wave path, this build"; python tools/time_wave_path.py 2>&1 | tee gpurun_out/r04/time_wave_path.log
echo "--- wave path, round-3 library"; SLS_HIP_LIB=$PWD/sequential-line-search_amd/libsls_hip_r03.so python tools/time_wave_path.py 2>&1 | tee gpurun_out/r04/time_wave_path_r03.log
