#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/c2ptrace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/c2ptrace -- python -u $R/tools/c2_predict_trace.py > /tmp/c2ptrace.log 2>&1
tail -1 /tmp/c2ptrace.log
f=$(find /tmp/c2ptrace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/c2_predict_timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slsk::", "")[:56]) for r in rows))
starts = [i for i, e in enumerate(ev) if "prep_kernel<true>" in e[2]]
a, b = starts[-3], starts[-2]
print(f"predict call: {(ev[b][0] - ev[a][0]) / 1e3:.1f} us from one prep_cands to the next")
prev_end = ev[a - 1][1]; busy = 0
for s, e, n in ev[a - 2:b]:
    print(f"  gap {(s - prev_end) / 1e3:8.1f} us   run {(e - s) / 1e3:8.1f} us   {n}"); busy += e - s; prev_end = max(prev_end, e)
print(f"  busy {busy / 1e3:.1f} us")
PY
