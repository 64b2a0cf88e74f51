#!/bin/bash
# Full chain with the fused FIR + correlation + envelope kernel against the band kernel + envelope kernel pair.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03/fce_tests.log 2>&1 || { tail -40 gpurun_out/r03/fce_tests.log; exit 1; }
tail -1 gpurun_out/r03/fce_tests.log
for fce in 0 1; do
  ECHO_FCE=$fce timeout -k 10 200 python bench.py --no-cpu-baseline --mode full > gpurun_out/r03/full_fce$fce.json 2> gpurun_out/r03/full_fce$fce.err || { tail gpurun_out/r03/full_fce$fce.err; exit 1; }
  ECHO_FCE=$fce timeout -k 10 200 python bench.py --no-cpu-baseline --mode full --lanes 1 > gpurun_out/r03/full_l1_fce$fce.json 2> gpurun_out/r03/full_l1_fce$fce.err || exit 1
  python tools/print_bench.py gpurun_out/r03/full_fce$fce.json gpurun_out/r03/full_l1_fce$fce.json
done
