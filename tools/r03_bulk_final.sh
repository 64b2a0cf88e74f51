#!/bin/bash
# Bulk parity of the optimized parameter set with the round's final library: both band sets, named and full chain.
mkdir -p gpurun_out/r03
timeout -k 10 1100 python -u tools/bulk_parity.py ${1:-200000} 16 gpurun_out/r03/bulk_parity_optimized_final.json 1 optimized > gpurun_out/r03/bulk_final.log 2>&1
echo "rc=$?"; grep -E '^\{' gpurun_out/r03/bulk_final.log | cut -c1-330
