#!/bin/bash
# Round-3 check run on the GPU box: the GPU test suite, then the bench lines kept under profiles/r03/.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03/gputests.log 2>&1 || { tail -30 gpurun_out/r03/gputests.log; exit 1; }
tail -3 gpurun_out/r03/gputests.log
timeout -k 10 300 python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err || { tail -20 gpurun_out/r03/bench_default.err; exit 1; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_driver_protocol.json 2> gpurun_out/r03/bench_driver_protocol.err || exit 1
timeout -k 10 200 python bench.py --lanes 1 --force-dist --no-cpu-baseline > gpurun_out/r03/bench_lanes1_force_dist.json 2> gpurun_out/r03/bench_lanes1_force_dist.err || { echo "lanes1 force-dist rc=$?"; tail -5 gpurun_out/r03/bench_lanes1_force_dist.err; exit 1; }
timeout -k 10 200 python bench.py --force-dist --no-cpu-baseline > gpurun_out/r03/bench_force_dist_one_rank.json 2> gpurun_out/r03/bench_force_dist_one_rank.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.3f M"%(d["value"]/1e6), "ms/step %.3f"%d["ms_per_step"], "frac %.4f"%d["roofline"]["frac"], d.get("steady_state",{}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
PY
