#!/bin/bash
# Whole GPU suite, smoke, then the default bench and the driver-protocol bench.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03/gputests.log 2>&1 || { tail -40 gpurun_out/r03/gputests.log; exit 1; }
tail -3 gpurun_out/r03/gputests.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r03/smoke.log 2>&1 || { tail -20 gpurun_out/r03/smoke.log; exit 1; }
tail -1 gpurun_out/r03/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err || { tail gpurun_out/r03/bench_default.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_protocol.json 2> gpurun_out/r03/bench_dp.err || exit 1
python - <<'PY'
import json
for f in ["gpurun_out/r03/bench_default.json","gpurun_out/r03/bench_driver_protocol.json"]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.3f M"%(d["value"]/1e6), "ms/step %.3f"%d["ms_per_step"], d.get("steady_state",{}).get("value"))
PY
