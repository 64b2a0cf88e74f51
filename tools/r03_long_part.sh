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
run orig_l3_own ECHO_IIR_SIDE_OWN=1 -- --workload original &&
run orig_l3_2048_own ECHO_IIR_SIDE_OWN=1 -- --workload original --frames 2048 &&
run orig_l3_8192_own ECHO_IIR_SIDE_OWN=1 -- --workload original --frames 8192 --steps 100 &&
run orig_l3_8192_own96 ECHO_IIR_SIDE_OWN=1 ECHO_IIR_CUS=96 -- --workload original --frames 8192 --steps 100 &&
run tp_l3_own ECHO_IIR_SIDE_OWN=1 -- --workload twopeak
