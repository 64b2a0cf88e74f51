#!/bin/bash
# Bulk parity of the full chain (envelope branch) with the round's library: both band sets against the oracle.
mkdir -p gpurun_out/r03
BULK_MODES=full timeout -k 10 1100 python -u tools/bulk_parity.py ${1:-300000} 16 gpurun_out/r03/bulk_parity_full_${1:-300000}.json 1 optimized > gpurun_out/r03/bulk_full.log 2>&1
echo "rc=$?"; grep -E '^\{' gpurun_out/r03/bulk_full.log | cut -c1-500
