#!/bin/bash
# Kernel timeline of one C3 submit (sequential_line_search_nd 32 30): rocprofv3 kernel trace -> gaps between the kernels of the
# last-but-one submit (from one map_opt_kernel to the next)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/c3trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/c3trace -- $R/sequential-line-search_amd/bin/sequential_line_search_nd 32 30 1 > /tmp/c3trace.log 2>&1
tail -2 /tmp/c3trace.log
f=$(find /tmp/c3trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $R/gpurun_out/c3_trace_gaps.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slsk::", "")[:56]) for r in rows))
starts = [i for i, e in enumerate(ev) if "map_opt_kernel" in e[2]]
a, b = starts[-3], starts[-2]
print(f"submit: {(ev[b][0] - ev[a][0]) / 1e3:.1f} us from one map_opt_kernel start to the next; {b - a} kernels")
prev_end = ev[a - 1][1]; busy = 0; gaps = 0
out = []
for s, e, n in ev[a:b]:
    out.append((s - prev_end, e - s, n)); busy += e - s; gaps += max(0, s - prev_end); prev_end = max(prev_end, e)
# compress runs of the same kernel name
i = 0
while i < len(out):
    j = i
    while j + 1 < len(out) and out[j + 1][2] == out[i][2]: j += 1
    n = j - i + 1
    print(f"  x{n:3d}  gap(sum) {sum(o[0] for o in out[i:j+1]) / 1e3:8.1f} us   run(sum) {sum(o[1] for o in out[i:j+1]) / 1e3:8.1f} us   {out[i][2]}")
    i = j + 1
print(f"  kernels busy {busy / 1e3:.1f} us, idle between kernels {gaps / 1e3:.1f} us")
PY
cat $R/gpurun_out/c3_trace_gaps.txt
