#!/bin/bash
# PMC passes over single variants of gemm_probe_ring (GPU box).  usage: pmc_gemm.sh "0 1021 104"
cd $GRAFT_REPO_ROOT/tools/probes; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS"
P3="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD"
P4="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
for v in ${1:-0 1021 104}; do
  n=1
  for P in "$P1" "$P2" "$P3" "$P4"; do
    d=$OUT/v${v}_p$n; rm -rf $d
    ONLY=$v timeout 300 rocprofv3 --kernel-trace --pmc $P -d $d -o r --output-format csv -- ./bin/gemm_probe_ring 16384 8192 8192 1 > $d.log 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$v" <<'PY'
import csv, sys, collections
f, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(f)):
        if "probe_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("variant", v, "no counters:", e); sys.exit(0)
# several dispatches of the kernel (warm-up + timed): the mean per dispatch
print("variant", v, " ".join(f"{k}={sum(x)/len(x):.4g}" for k, x in sorted(acc.items())))
PY
    n=$((n+1))
  done
done
