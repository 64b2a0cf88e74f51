#!/bin/bash
# Whole GPU suite, then the long-frame bench lines.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03/gputests.log 2>&1 || { tail -40 gpurun_out/r03/gputests.log; exit 1; }
tail -3 gpurun_out/r03/gputests.log
for w in original twopeak; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w > gpurun_out/r03/bench_${w}_variant.json 2> gpurun_out/r03/bench_${w}.err || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 5 > gpurun_out/r03/bench_${w}_variant_driver_protocol.json 2> gpurun_out/r03/bench_${w}_dp.err || exit 1
done
timeout -k 10 200 python bench.py --no-cpu-baseline --mode full --lanes 1 > gpurun_out/r03/bench_full_mode.json 2> gpurun_out/r03/bench_full.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/bench_*variant*.json"))+["gpurun_out/r03/bench_full_mode.json"]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "%.3f M"%(d["value"]/1e6), "ms/step %.3f"%d["ms_per_step"], d.get("steady_state",{}).get("value"))
PY
