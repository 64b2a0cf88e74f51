#!/bin/bash
# Bulk parity on a second range of synthetic frames (from index 1,000,000): the named chain on three lanes, then the
# full chain, both band sets each.
mkdir -p gpurun_out/r03
BULK_START=1000000 BULK_MODES=named timeout -k 10 1100 python -u tools/bulk_parity.py ${1:-1000000} 16 gpurun_out/r03/bulk_parity_named_range2.json 3 optimized > gpurun_out/r03/bulk_range2_named.log 2>&1
rc=$?; echo "named rc=$rc"; grep -E '^\{' gpurun_out/r03/bulk_range2_named.log | cut -c1-330
[ $rc -eq 0 ] || exit $rc
BULK_START=1000000 BULK_MODES=full timeout -k 10 1100 python -u tools/bulk_parity.py ${2:-400000} 16 gpurun_out/r03/bulk_parity_full_range2.json 1 optimized > gpurun_out/r03/bulk_range2_full.log 2>&1
rc=$?; echo "full rc=$rc"; grep -E '^\{' gpurun_out/r03/bulk_range2_full.log | cut -c1-330
exit $rc
