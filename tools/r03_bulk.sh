#!/bin/bash
# Bulk parity of every parameter set with the round's final library (tools/bulk_parity.py): 200,000 frames per leg.
mkdir -p gpurun_out/r03
timeout -k 10 200 python -u -m pytest tests/test_gpu_lanes.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for v in optimized original twopeak realtime; do
  echo "== $v"
  timeout -k 10 700 python -u tools/bulk_parity.py 200000 16 gpurun_out/r03/bulk_parity_${v}_200k.json 1 $v 2>&1 | grep -E '^\{|Error|error' | cut -c1-400
  echo "rc=${PIPESTATUS[0]}"
done
echo "== optimized, three lanes"
timeout -k 10 700 python -u tools/bulk_parity.py 200000 16 gpurun_out/r03/bulk_parity_optimized_200k_lanes3.json 3 optimized 2>&1 | grep -E '^\{|Error|error' | cut -c1-300
echo "rc=${PIPESTATUS[0]}"
