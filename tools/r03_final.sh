#!/bin/bash
# End-of-round records: whole GPU suite, smoke, the bench records of tools/r03_records.sh, the full chain's profile set.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03/gputests.log 2>&1 || { tail -40 gpurun_out/r03/gputests.log; exit 1; }
tail -1 gpurun_out/r03/gputests.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r03/smoke.log 2>&1 || { tail -20 gpurun_out/r03/smoke.log; exit 1; }
tail -1 gpurun_out/r03/smoke.log
bash tools/r03_records.sh || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_protocol.json 2> gpurun_out/r03/bench_dp.err || exit 1
python tools/print_bench.py gpurun_out/r03/bench_default.json gpurun_out/r03/bench_driver_protocol.json
bash tools/collect_profiles.sh gpurun_out/r03/prof_full --mode full > gpurun_out/r03/prof_full.log 2>&1 || { tail -20 gpurun_out/r03/prof_full.log; exit 1; }
tail -8 gpurun_out/r03/prof_full.log
