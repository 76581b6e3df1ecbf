#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_host_cpp.py -q -x 2>&1 | tail -12
B=sequential-line-search_amd/bin
for i in 1 2 3; do SLS_HOST_TIMING=1 $B/sequential_line_search_nd 32 30 1 > gpurun_out/r04/c3_run$i.log 2>&1; done
python - <<'PY'
import re,statistics
for f in ("c3_run1","c3_run2","c3_run3"):
    t=open(f"gpurun_out/r04/{f}.log").read()
    ms=[float(v) for v in re.findall(r" ms ([-\d.e]+)",t)]
    fit=[float(v) for v in re.findall(r"MAP fit ([\d.]+) ms",t)]; nx=[float(v) for v in re.findall(r"next point ([\d.]+) ms",t)]
    loc=[float(v) for v in re.findall(r"local phase ([\d.]+) ms",t)]
    print(f,"mean w/o first",statistics.mean(ms[1:]),"median",statistics.median(ms),"max",max(ms[1:]),"map fit mean",statistics.mean(fit[1:]),"next point mean",statistics.mean(nx[1:]),"local phase mean",statistics.mean(loc[1:]))
PY
SLS_WAVE_TRACE=1 $B/sequential_line_search_nd 32 30 1 2>&1 | grep "wave trace" | tail -3
SLS_WAVE_COOP=0 SLS_WAVE_TRACE=1 $B/sequential_line_search_nd 32 30 1 2>&1 | grep "wave trace" | tail -2
python tools/time_wave_path.py 2>&1 | tee gpurun_out/r04/time_wave_path.log
