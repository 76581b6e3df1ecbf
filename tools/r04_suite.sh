#!/bin/bash
# round 4: whole GPU suite + C3 / C1 timings (no CPU legs: tools/run_configs.py has them)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04/pytest_gpu.log; tail -22 gpurun_out/r04/pytest_gpu.log | cut -c1-250
cp gpurun_out/c5_full_fit.json gpurun_out/map_optima_report.json gpurun_out/test_evidence.json gpurun_out/r04/ 2>/dev/null
B=sequential-line-search_amd/bin
for i in 1 2; do SLS_HOST_TIMING=1 $B/sequential_line_search_nd 32 30 1 > gpurun_out/r04/c3_run$i.log 2>&1; done
python - <<'PY'
import re,statistics
for f in ("c3_run1","c3_run2"):
    t=open(f"gpurun_out/r04/{f}.log").read()
    ms=[float(v) for v in re.findall(r" ms ([-\d.e]+)",t)]
    fit=[float(v) for v in re.findall(r"MAP fit ([\d.]+) ms",t)]; nx=[float(v) for v in re.findall(r"next point ([\d.]+) ms",t)]
    print(f,"mean w/o first",statistics.mean(ms[1:]),"median",statistics.median(ms),"max",max(ms[1:]),"map fit mean",statistics.mean(fit[1:]),"next point mean",statistics.mean(nx[1:]))
PY
python tools/time_wave_path.py 2>&1 | tee gpurun_out/r04/time_wave_path.log
SLS_MAP_TRACE=1 SLS_WAVE_TRACE=1 SLS_HOST_TIMING=1 $B/sequential_line_search_nd 32 30 1 > gpurun_out/r04/c3_trace.log 2>&1; grep "wave trace\|map_opt trace" gpurun_out/r04/c3_trace.log | tail -4
python tools/time_map_fit.py 2>&1 | grep "one launch" | tee gpurun_out/r04/time_map_fit.log
