#!/bin/bash
# round 4: the device-resident MAP fits -- tests + C3 / C1 timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_map_device.py -q > gpurun_out/r04/map_device.log 2>&1; echo "exit $?" >> gpurun_out/r04/map_device.log; tail -25 gpurun_out/r04/map_device.log
timeout 900 python -m pytest tests/test_gpu_map_fit.py tests/test_gpu_host_cpp.py -x -q -k "not c5_full" > gpurun_out/r04/map_fit.log 2>&1; echo "exit $?" >> gpurun_out/r04/map_fit.log; tail -4 gpurun_out/r04/map_fit.log
python tools/time_map_fit.py 2>&1 | tee gpurun_out/r04/time_map_fit.log
B=sequential-line-search_amd/bin
for i in 1 2; do SLS_HOST_TIMING=1 $B/sequential_line_search_nd 32 30 1 > gpurun_out/r04/c3_run$i.log 2>&1; done
SLS_WAVE_TRACE=1 SLS_HOST_TIMING=1 $B/sequential_line_search_nd 32 30 1 > gpurun_out/r04/c3_trace.log 2>&1; grep "wave trace" gpurun_out/r04/c3_trace.log | tail -8
python - <<'PY'
import re,statistics
for f in ("c3_run1","c3_run2"):
    t=open(f"gpurun_out/r04/{f}.log").read()
    ms=[float(v) for v in re.findall(r" ms ([-\d.e]+)",t)]
    fit=[float(v) for v in re.findall(r"MAP fit ([\d.]+) ms",t)]; nx=[float(v) for v in re.findall(r"next point ([\d.]+) ms",t)]
    print(f,"mean w/o first",statistics.mean(ms[1:]),"median",statistics.median(ms),"map fit mean",statistics.mean(fit[1:]),"next point mean",statistics.mean(nx[1:]))
PY
for i in 1 2 3; do s=$(date +%s.%N); $B/bayesian_optimization_1d 1 20 1 | tail -1; e=$(date +%s.%N); echo "C1 wall $(echo "$e - $s" | bc) s"; done
s=$(date +%s.%N); SLS_MAP_DEVICE=0 $B/bayesian_optimization_1d 1 20 1 | tail -1; e=$(date +%s.%N); echo "C1 host-driven wall $(echo "$e - $s" | bc) s"
