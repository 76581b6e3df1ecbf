#!/bin/bash
# rocprofv3 PMC evidence for profiles/r04b_pmc_*: separate passes (as the MI355X guide prescribes; --kernel-trace only beside --pmc)
# of (a) the full 65 536-candidate launch shape of the bench (SLS_COMPACT=0: every acq_gemm dispatch has the same shape) and
# (b) GP fits at N = 2048 / 4096 (the fused factor + inverse launch: MFMA pipe busy of a chain-bound kernel)
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r04b
mkdir -p $OUT $R/gpurun_out/r04b
cd /tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --n-local 12 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
export SLS_COMPACT=0
BENCH="python $R/bench.py --steps 1 --warmup 1 --n-local 6 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -d $OUT/pmc_mfma -o pmc -- $BENCH > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
unset SLS_COMPACT
python $R/tools/summarize_prof.py $OUT $R/gpurun_out/r04b/r04b > $OUT/summary.log 2>&1; tail -12 $OUT/summary.log
cd $R && python tools/make_pmc_summary.py gpurun_out/r04b/r04b >> $OUT/summary.log 2>&1; tail -3 $OUT/summary.log
# (b) the fits
cat > /tmp/fits.py <<'PY'
import importlib, os, sys
import numpy as np
R = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, R)
sls = importlib.import_module("sequential-line-search_amd")
rng = np.random.default_rng(1)
ctx = sls.Context(0)
for N in (2048, 4096):
    X = np.asfortranarray(rng.uniform(0, 1, (16, N))); y = np.sin(X.sum(axis=0)) + 0.01 * rng.standard_normal(N)
    th = np.concatenate([[0.5], np.full(16, 0.7)])
    for _ in range(4): sls.GP(ctx, X, y, th, 0.005, 0).close()
PY
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 -d $OUT/fit_mfma -o pmc --output-format csv -- python /tmp/fits.py > $OUT/fit_mfma.log 2>&1
f=$(find $OUT/fit_mfma -name "*counter_collection.csv" | head -1)
python3 - "$f" $R/gpurun_out/r04b/r04b_pmc_fused_fit.json <<'PY'
import csv, sys, json, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "potrf_dataflow_kernel" not in r["Kernel_Name"]: continue
    acc[(r["Grid_Size"], r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
per = collections.defaultdict(list)
for (grid, disp), c in acc.items():
    d = {k: sum(v) for k, v in c.items()}
    per[grid].append(d)
out = {}
for grid, ds in per.items():
    ds = ds[1:] if len(ds) > 1 else ds          # first dispatch: cold
    m = {k: sum(d[k] for d in ds) / len(ds) for k in ds[0]}
    gui = m["GRBM_GUI_ACTIVE"] / 8
    out[f"grid_{grid}"] = {"dispatches": len(ds), **m, "mfma_flops": m["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512,
                           "mfma_pipe_busy_fraction_of_the_chip": m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / gui}
json.dump(out, open(sys.argv[2], "w"), indent=1); print(json.dumps(out, indent=1)[:1500])
PY
