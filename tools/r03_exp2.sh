#!/bin/bash
# Experiments: IIR mappings under three lanes
source <(sed -n '/^run() {/,/^}/p' tools/r03_exp1.sh)
mkdir -p gpurun_out/r03/exp
run base X=1 -- || exit 1
run iirmode1_l3 ECHO_IIR_MODE=1 -- || exit 1
run iirmode1_l4 ECHO_IIR_MODE=1 -- --lanes 4 || exit 1
run iirmode1_l2 ECHO_IIR_MODE=1 -- --lanes 2 || exit 1
run lps16_l3 ECHO_IIR_LPS=16 -- || exit 1
run iirmode1_q8_l5 ECHO_IIR_MODE=1 GPU_MAX_HW_QUEUES=8 -- --lanes 5 || exit 1
