#!/bin/bash
# new drop-in tests, then the one-stream (prefetch) bench with and without a one-rank RCCL group
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 300 python -u -m pytest tests/test_gpu_dropin.py tests/test_gpu_threads.py "tests/test_gpu_parity.py::test_dropin_function_surface" -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -15 || exit 1
echo "== lanes 1"
timeout -k 10 120 python bench.py --lanes 1 --no-cpu-baseline > gpurun_out/r03/bench_one_stream_prefetch.json 2> gpurun_out/r03/bench_one_stream_prefetch.err || { echo "lanes1 rc=$?"; exit 1; }
echo "== lanes 1 force-dist"
timeout -k 10 100 python bench.py --lanes 1 --force-dist --no-cpu-baseline > gpurun_out/r03/bench_lanes1_force_dist.json 2> gpurun_out/r03/bench_lanes1_force_dist.err || { echo "lanes1 force-dist rc=$?"; tail -5 gpurun_out/r03/bench_lanes1_force_dist.err; exit 1; }
echo "== lanes 3 force-dist"
timeout -k 10 100 python bench.py --force-dist --no-cpu-baseline > gpurun_out/r03/bench_force_dist_one_rank.json 2> gpurun_out/r03/bench_force_dist_one_rank.err || { echo "force-dist rc=$?"; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.3f M"%(d["value"]/1e6), "ms/step %.3f"%d["ms_per_step"], "frac %.4f"%d["roofline"]["frac"], d.get("steady_state",{}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
PY
