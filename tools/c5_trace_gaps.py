"""Idle time before each kernel of one C5 evaluation, from a rocprofv3 kernel-trace csv: the last complete evaluation before the
final one (an evaluation = from one prep_kernel to the next)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows))
starts = [i for i, e in enumerate(ev) if e[2].startswith("void slsk::prep_kernel") or "prep_kernel" in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3      # which prep from the end starts the window
a, b = starts[-k], starts[-k + 1]
print(f"evaluation: {(ev[b][0] - ev[a][0]) / 1e3:.1f} us from prep start to the next prep start")
prev_end = ev[a - 1][1]
busy = 0
for s, e, n in ev[a:b]:
    print(f"  gap {(s - prev_end) / 1e3:8.1f} us   run {(e - s) / 1e3:8.1f} us   {n}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"  kernels busy {busy / 1e3:.1f} us; last kernel end -> next prep start {(ev[b][0] - prev_end) / 1e3:.1f} us")
