#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ubench
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/f -o pmc -- $R/tools/probes/bin/ubench > $OUT/log.txt 2>&1
python - <<PY
import sqlite3
c = sqlite3.connect("$OUT/f/pmc_results.db")
from collections import defaultdict
per = defaultdict(float); order = []
for disp, kn, cn, val, gx in c.execute("select dispatch_id,kernel_name,counter_name,value,grid_size from counters_collection"):
    if "gemm_kernel" not in kn: continue
    if (disp, kn, gx) not in order: order.append((disp, kn, gx))
    per[(disp, cn)] += val
for disp, kn, gx in order:
    print(disp, kn[:40], "grid", gx, "FETCH_KB %.4g" % per[(disp, "FETCH_SIZE")], "hit %.4g miss %.4g" % (per[(disp, "TCC_HIT_sum")], per[(disp, "TCC_MISS_sum")]))
PY
