#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
bash tools/r04_c3.sh 2>&1 | grep -v "^---" | head -40
tools/probes/bin/vendor_yardstick > gpurun_out/r04/vendor_yardstick.json 2> gpurun_out/r04/vendor_yardstick.err; cat gpurun_out/r04/vendor_yardstick.json; tail -3 gpurun_out/r04/vendor_yardstick.err
timeout 900 python tools/run_configs.py > gpurun_out/r04/run_configs.log 2>&1; tail -60 gpurun_out/r04/run_configs.log
cp gpurun_out/configs.json gpurun_out/r04/configs.json
