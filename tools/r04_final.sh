#!/bin/bash
# round 4, final state: whole GPU suite, C3 runs, evidence (bench line + rocprof), configs with CPU legs
cd $GRAFT_REPO_ROOT
bash tools/r04_suite.sh
bash tools/r04_evidence.sh
timeout 1200 python tools/run_configs.py > gpurun_out/r04/run_configs.log 2> gpurun_out/r04/run_configs.err; grep "run_configs" gpurun_out/r04/run_configs.err; cp gpurun_out/configs.json gpurun_out/r04/configs.json
