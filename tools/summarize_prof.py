#!/usr/bin/env python3
"""Turn the rocprofv3 rocpd databases under gpurun_out/<dir> into the small committed summaries in profiles/.
usage: tools/summarize_prof.py gpurun_out/prof_r01 profiles/r01"""
import json
import os
import sqlite3
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst), exist_ok=True)


def short(n):
    n = n.replace("void ", "").replace("slsk::", "")
    return n.split("(")[0]


# 1. kernel-trace stats
db = sqlite3.connect(os.path.join(src, "trace", "trace_results.db"))
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(dst + "_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_us,avg_us,percent\n")
    for n, c, t, a, p in rows:
        f.write(f"\"{short(n)}\",{c},{t:.1f},{a:.2f},{p:.3f}\n")
print(open(dst + "_kernel_stats.csv").read())

# 2. PMC per kernel (mean over dispatches; counters are summed over their instances per dispatch)
pmc = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_mfma", "pmc_fetch", "pmc_write", "pmc_sq"):
    p = os.path.join(src, sub, "pmc_results.db")
    if not os.path.exists(p):
        continue
    d = sqlite3.connect(p)
    per = defaultdict(float)
    dur = {}
    for disp, kn, cn, val, st, en in d.execute(
            "select dispatch_id,kernel_name,counter_name,value,start,end from counters_collection"):
        per[(disp, kn, cn)] += val
        dur[(disp, kn)] = en - st
    for (disp, kn, cn), v in per.items():
        pmc[short(kn)][cn].append(v)
    for (disp, kn), v in dur.items():
        pmc[short(kn)]["duration_ns[" + sub + "]"].append(v)
summary = {}
for kn, cs in pmc.items():
    summary[kn] = {cn: {"mean": sum(v) / len(v), "n": len(v)} for cn, v in cs.items()}
with open(dst + "_pmc_by_kernel.json", "w") as f:
    json.dump(summary, f, indent=1, sort_keys=True)
for kn in summary:
    if "acq_gemm" in kn or "grad_gemm" in kn or "cross_gram" in kn:
        print(kn)
        for cn, v in sorted(summary[kn].items()):
            print(f"   {cn:40s} {v['mean']:.6g}  (n={v['n']})")
