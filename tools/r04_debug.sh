#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
P=sequential-line-search_amd
cp $P/libsls_hip.so /tmp/libsls_hip_dpp.so
for v in dpp; do
  [ $v = shfl ] && cp $P/libsls_hip_shfl.so $P/libsls_hip.so
  echo "=== variant $v"
  timeout 300 python -m pytest tests/test_gpu_map_device.py -q -x -k "5-1-3-False-0" 2>&1 | grep -E "assert|Error|passed|failed|fault" | head -12
  timeout 300 python -m pytest tests/test_gpu_map_device.py -q -x -k "40-8-30-False-0" 2>&1 | grep -E "assert|Error|passed|failed|fault" | head -12
  timeout 600 python -m pytest tests/test_gpu_map_device.py -q 2>&1 | tail -15
  timeout 100 $P/bin/bayesian_optimization_1d 1 20 1 2>&1 | tail -2
done
cp /tmp/libsls_hip_dpp.so $P/libsls_hip.so
