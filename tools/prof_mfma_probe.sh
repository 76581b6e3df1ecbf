#!/bin/bash
# effective clock and MFMA-pipe occupancy of the register-only MFMA loops (csrc/mfma_probe)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_mfma
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d $OUT -o pmc -- $R/tools/probes/bin/mfma_probe > $OUT/log.txt 2>&1
python - <<PY
import sqlite3, glob
from collections import defaultdict
c = sqlite3.connect(glob.glob("$OUT/*.db")[0])
acc = defaultdict(float); dur = {}; name = {}
for disp, kn, cn, val, s, e in c.execute("select dispatch_id,kernel_name,counter_name,value,start,end from counters_collection"):
    acc[(disp, cn)] += val; dur[disp] = (e - s); name[disp] = kn
for d in sorted(dur):
    gui = acc[(d, "GRBM_GUI_ACTIVE")] / 8; busy = acc[(d, "SQ_VALU_MFMA_BUSY_CYCLES")] / 1024; n = acc[(d, "SQ_INSTS_VALU_MFMA_MOPS_F64")] * 512 / 2048
    if dur[d] < 1e6: continue
    print(name[d][:40], "dur %.2f ms  clock %.2f GHz  MFMA busy %.3f of active  busy cycles/MFMA %.1f" % (dur[d] / 1e6, gui / dur[d], busy / gui, acc[(d, "SQ_VALU_MFMA_BUSY_CYCLES")] / max(n, 1)))
PY
