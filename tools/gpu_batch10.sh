#!/bin/bash
# acq_gemm A/B: wave priority for the second workgroup of every CU, gate phase 0 / 2000, ungated
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --n-local 8 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms/step', round(j['ms_per_step'],1), 'acq_gemm frac', round(j['roofline']['frac'],4), 'acq ms', round(j['stage_ms_per_step']['acq_gemm'],1))"; }
run SLS_ACQ_PRIO=0
run SLS_ACQ_PRIO=1
run SLS_ACQ_PRIO=1 SLS_GATE_PHASE=0
run SLS_ACQ_PRIO=0 SLS_GATE_PHASE=0
run SLS_ACQ_PRIO=0 SLS_PERSIST=0
run SLS_ACQ_PRIO=0
run SLS_ACQ_PRIO=1
