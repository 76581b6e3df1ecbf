#!/bin/bash
# chain_gemm with bare barriers: potrf parity tests, potrf_bench with trace, fit stages; FETCH_SIZE of acq_gemm at gate phase 0
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "potrf or chol or fit or gp_" 2>&1 | tail -3
cd tools/probes && POTRF_BENCH_QUICK=1 POTRF_BENCH_TRACE=1 timeout 200 ./bin/potrf_bench 2048 4096 8192 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r02/potrf_chainfix.log | grep -v "^  *[0-9]* |" ; grep "^  *[0-9]* |" $GRAFT_REPO_ROOT/gpurun_out/r02/potrf_chainfix.log | tail -12
cd /tmp; export TMPDIR=/tmp
for ph in 0 2000; do
  rm -rf /tmp/pf$ph
  SLS_COMPACT=0 SLS_GATE_PHASE=$ph rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf$ph -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --n-local 6 --no-cpu-baseline > /tmp/pf$ph.log 2>&1
  python3 - /tmp/pf$ph $ph <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "acq_gemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("gate phase", sys.argv[2], "acq_gemm launches", len(v), "FETCH_SIZE mean KB", sum(v)/len(v), "-> GB x2:", 2*sum(v)/len(v)*1024/1e9)
PY
done
