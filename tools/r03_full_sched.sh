#!/bin/bash
# Full chain with the fused kernel: how to overlap its batches (lanes, where the IIR runs).
set -o pipefail
mkdir -p gpurun_out/r03/exp
run() { name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --mode full $ARGS > gpurun_out/r03/exp/fullfce_$name.json 2> gpurun_out/r03/exp/fullfce_$name.err || { tail gpurun_out/r03/exp/fullfce_$name.err; exit 1; }
  python tools/print_bench.py gpurun_out/r03/exp/fullfce_$name.json
}
ARGS="--lanes 3" run l3_default X=1 &&
ARGS="--lanes 3" run l3_inline ECHO_IIR_SIDE=0 &&
ARGS="--lanes 3" run l3_cu64 ECHO_IIR_CUS=64 &&
ARGS="--lanes 3" run l3_cu96 ECHO_IIR_CUS=96 &&
ARGS="--lanes 3" run l3_cu128 ECHO_IIR_CUS=128 &&
ARGS="--lanes 2" run l2_default X=1 &&
ARGS="--lanes 2" run l2_inline ECHO_IIR_SIDE=0 &&
ARGS="--lanes 1" run l1 X=1 &&
ARGS="--no-pipeline" run serial X=1 &&
ARGS="--lanes 3 --steps 20 --warmup 5" run l3_dp X=1 &&
ARGS="--lanes 1 --steps 20 --warmup 5" run l1_dp X=1
