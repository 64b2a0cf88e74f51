#!/bin/bash
# Long-frame sets with the matched filter as one real transform: parity tests, then bench lines with it and without.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q --timeout 180 --timeout-method thread -k "long or original or twopeak or mad or disputed or variant" > gpurun_out/r03/rcf_tests.log 2>&1 || { tail -40 gpurun_out/r03/rcf_tests.log; exit 1; }
tail -3 gpurun_out/r03/rcf_tests.log
run() { # name, env..., -- args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/rcf_$name.json 2> gpurun_out/r03/rcf_$name.err || { echo "$name failed"; tail -5 gpurun_out/r03/rcf_$name.err; return 1; }
    python - "$name" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03/rcf_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.3f M"%(d["value"]/1e6), "ms/step %.3f"%d["ms_per_step"], d.get("steady_state",{}).get("value"))
PY
}
run original_rcf X=1 -- --workload original &&
run original_blocks ECHO_RCF=0 -- --workload original &&
run twopeak_rcf X=1 -- --workload twopeak &&
run twopeak_blocks ECHO_RCF=0 -- --workload twopeak
