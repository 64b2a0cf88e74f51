#!/bin/bash
source <(sed -n '/^run() {/,/^}/p' tools/r03_exp1.sh)
mkdir -p gpurun_out/r03/exp
run orig_l1 X=1 -- --workload original --lanes 1 || exit 1
run orig_l1_gate2 ECHO_PF_GATE=2 -- --workload original --lanes 1 || exit 1
run twopeak_l1_gate2 ECHO_PF_GATE=2 -- --workload twopeak --lanes 1 || exit 1
run full_l1_gate2 ECHO_PF_GATE=2 -- --mode full --lanes 1 || exit 1
