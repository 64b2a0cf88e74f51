#!/bin/bash
# airSonar.py set: the IIR confined to a subset of the compute units (it stacks there), the other kernels free.
set -o pipefail
mkdir -p gpurun_out/r03/part
run() { # name, env..., -- args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/part/$name.json 2> gpurun_out/r03/part/$name.err || { echo "$name failed"; tail -5 gpurun_out/r03/part/$name.err; return 1; }
    python - "$name" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03/part/%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "%.3f M"%(d["value"]/1e6), "ms/step %.3f"%d["ms_per_step"], {k: round(v,3) for k,v in d["kernel_ms"].items()})
PY
}
run named_l3 X=1 -- &&
run full_l3 X=1 -- --mode full &&
run full_l3_forced ECHO_IIR_SIDE=1 ECHO_IIR_CUS=32 -- --mode full &&
run full_l3_off ECHO_IIR_SIDE=0 -- --mode full &&
run full_l1 X=1 -- --mode full --lanes 1 &&
run orig_l3 X=1 -- --workload original
