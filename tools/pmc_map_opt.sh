#!/bin/bash
# Hardware counters of map_opt_kernel over one run of config C3 (sequential_line_search_nd 32 30, MAP hyper-parameters on): instruction
# mix, matrix-core busy cycles, LDS bank conflicts and wait cycles of the one-workgroup kernel.  Counter passes only beside
# --kernel-trace.   gpurun -- 'bash tools/pmc_map_opt.sh'
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_map_opt; mkdir -p $O
cd /tmp
pass() {
  n=$1; shift
  rm -rf /tmp/pmc_$n; mkdir -p /tmp/pmc_$n
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$n -o p --output-format csv -- $R/sequential-line-search_amd/bin/sequential_line_search_nd 32 30 1 1 > /tmp/pmc_$n/log.txt 2>&1
  python3 - "$(find /tmp/pmc_$n -name '*counter_collection.csv' | head -1)" <<'PY'
import collections, csv, sys
tot = collections.defaultdict(float); n = collections.defaultdict(set); ns = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "map_opt_kernel" not in r["Kernel_Name"]: continue
    tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
    ns[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("dispatches", len(ns), "total ms", sum(ns.values()) / 1e6)
for k in sorted(tot): print(f"  {k:36s} {tot[k]:16.0f}  per us {tot[k] / (sum(ns.values()) / 1e3):10.1f}")
PY
}
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT 2>&1 | tee $O/pass1.txt
pass 2 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT 2>&1 | tee $O/pass2.txt
pass 3 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_INSTS_BRANCH 2>&1 | tee $O/pass3.txt
tail -5 /tmp/pmc_3/log.txt
