#!/bin/bash
# A/B of tile scheduling variants inside the real acq_gemm_kernel: FETCH_SIZE and TCC hit/miss passes.
# usage: prof_gate_variants.sh "SLS_PERSIST=0 SLS_GATE_PHASE=0" "SLS_PERSIST=1 SLS_GATE_PHASE=2000" ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_gate_variants
rm -rf $OUT; mkdir -p $OUT
cd /tmp
BENCH="python $R/bench.py --steps 1 --warmup 0 --n-local 3 --no-cpu-baseline"
i=0
for cfg in "$@"; do
  i=$((i+1)); st=$i
  env $cfg timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/s$st -o pmcf -- $BENCH > $OUT/log_f$st.txt 2>&1
  env $cfg timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/s$st -o pmch -- $BENCH > $OUT/log_h$st.txt 2>&1
  python - <<PY
import sqlite3, glob
from collections import defaultdict
per = defaultdict(list)
for f in glob.glob("$OUT/s$st/*.db"):
    c = sqlite3.connect(f)
    acc = defaultdict(float); dur = {}
    for disp, kn, cn, val, s, e in c.execute("select dispatch_id,kernel_name,counter_name,value,start,end from counters_collection"):
        if "acq_gemm_kernel" not in kn: continue
        acc[(disp, cn)] += val; dur[disp] = (e - s) / 1e6
    for (disp, cn), v in acc.items(): per[cn].append(v)
    per["dur"] += list(dur.values())
import statistics as S
m = lambda k: S.mean(per[k]) if per[k] else float("nan")
print("$cfg launches", len(per["FETCH_SIZE"]), "ms %.2f" % m("dur"), "FETCH_GB(raw) %.2f (x2 = %.1f GB)" % (m("FETCH_SIZE") * 1024 / 1e9, 2 * m("FETCH_SIZE") * 1024 / 1e9),
      "hitrate %.3f" % (m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum"))),
      "min/max FETCH raw GB %.1f/%.1f" % (min(per["FETCH_SIZE"]) * 1024 / 1e9, max(per["FETCH_SIZE"]) * 1024 / 1e9))
PY
done
