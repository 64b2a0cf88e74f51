#!/bin/bash
# Full chain on one stream with the next batch's IIR prefetched: when may the prefetched IIR start?
set -o pipefail
mkdir -p gpurun_out/r03/exp
for g in 1 2 0; do
  ECHO_PF_GATE=$g timeout -k 10 200 python bench.py --no-cpu-baseline --mode full --lanes 1 > gpurun_out/r03/exp/fullfce_l1_gate$g.json 2> gpurun_out/r03/exp/fullfce_l1_gate$g.err || { tail gpurun_out/r03/exp/fullfce_l1_gate$g.err; exit 1; }
  python tools/print_bench.py gpurun_out/r03/exp/fullfce_l1_gate$g.json
done
ECHO_PF_GATE=2 timeout -k 10 200 python bench.py --no-cpu-baseline --mode full --lanes 1 --steps 20 --warmup 5 > gpurun_out/r03/exp/fullfce_l1_gate2_dp.json 2>/dev/null && python tools/print_bench.py gpurun_out/r03/exp/fullfce_l1_gate2_dp.json
