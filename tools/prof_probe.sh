#!/bin/bash
# usage: prof_probe.sh "M N K pad gm" ...   -> FETCH_SIZE / TCC hit+miss of the second launch of each configuration
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_probe
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  timeout 40 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p$i -o pmcf -- $R/tools/probes/bin/gemm_probe $cfg > $OUT/log$i.txt 2>&1
  timeout 40 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/p$i -o pmch -- $R/tools/probes/bin/gemm_probe $cfg >> $OUT/log$i.txt 2>&1
  python - <<PY
import sqlite3, glob
from collections import defaultdict
per = defaultdict(float); last = 0
for f in glob.glob("$OUT/p$i/*.db"):
    c = sqlite3.connect(f)
    for disp, kn, cn, val, st, en in c.execute("select dispatch_id,kernel_name,counter_name,value,start,end from counters_collection"):
        if "gemm_kernel" not in kn and "probe_kernel" not in kn: continue
        per[(disp, cn)] += val; per[(disp, "dur")] = (en - st) / 1e6; last = max(last, disp)
if True:
    print("$cfg".ljust(28), "ms %.2f" % per[(last, "dur")], "FETCH_GB(raw) %.2f" % (per[(last, "FETCH_SIZE")] * 1024 / 1e9),
          "hit %.3g miss %.3g hitrate %.3f" % (per[(last, "TCC_HIT_sum")], per[(last, "TCC_MISS_sum")],
          per[(last, "TCC_HIT_sum")] / max(1.0, per[(last, "TCC_HIT_sum")] + per[(last, "TCC_MISS_sum")])))
PY
done
