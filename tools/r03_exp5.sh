#!/bin/bash
# long-frame spectra with in-register butterflies: parity tests of the long-frame sets, then the bench of both
source <(sed -n '/^run() {/,/^}/p' tools/r03_exp1.sh)
mkdir -p gpurun_out/r03/exp
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q --timeout 120 --timeout-method thread -k "original or twopeak or odd or other_frame or long or mad or spectrum or non_finite" 2>&1 | tail -4 || exit 1
run orig_l3_bfly X=1 -- --workload original || exit 1
run orig_serial_bfly X=1 -- --workload original --no-pipeline || exit 1
run twopeak_l1_bfly X=1 -- --workload twopeak --lanes 1 || exit 1
