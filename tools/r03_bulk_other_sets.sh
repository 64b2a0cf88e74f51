#!/bin/bash
# Bulk parity of the other parameter sets with the round's final library: the two long-frame sets with three batches in
# flight, the realtime set on one stream; 200,000 frames each, the default frame range (so that the known disputed frames
# of the long-chirp sets are the ones of tests/golden/disputed_long_chirp.npz).  The long-chirp sets exit non-zero
# because of those frames; the summary lines tell.
mkdir -p gpurun_out/r03
for v in original twopeak; do
  timeout -k 10 500 python -u tools/bulk_parity.py ${1:-200000} 16 gpurun_out/r03/bulk_parity_${v}_final.json 3 $v > gpurun_out/r03/bulk_${v}_final.log 2>&1
  echo "$v rc=$?"; grep -E '^\{' gpurun_out/r03/bulk_${v}_final.log | cut -c1-420
done
timeout -k 10 500 python -u tools/bulk_parity.py ${1:-200000} 16 gpurun_out/r03/bulk_parity_realtime_final.json 1 realtime > gpurun_out/r03/bulk_realtime_final.log 2>&1
echo "realtime rc=$?"; grep -E '^\{' gpurun_out/r03/bulk_realtime_final.log | cut -c1-420
