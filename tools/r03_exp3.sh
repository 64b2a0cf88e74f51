#!/bin/bash
# config 4 (airSonar.py parameters) and the full chain: how the batches are overlapped
source <(sed -n '/^run() {/,/^}/p' tools/r03_exp1.sh)
mkdir -p gpurun_out/r03/exp
run orig_l3 X=1 -- --workload original || exit 1
run orig_l1 X=1 -- --workload original --lanes 1 || exit 1
run orig_l1_gate0 ECHO_PF_GATE=0 -- --workload original --lanes 1 || exit 1
run orig_serial X=1 -- --workload original --no-pipeline || exit 1
run orig_l2 X=1 -- --workload original --lanes 2 || exit 1
run full_l1 X=1 -- --mode full --lanes 1 || exit 1
run full_l3 X=1 -- --mode full || exit 1
run full_serial X=1 -- --mode full --no-pipeline || exit 1
run twopeak_l1 X=1 -- --workload twopeak --lanes 1 || exit 1
run twopeak_l3 X=1 -- --workload twopeak || exit 1
