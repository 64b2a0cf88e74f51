#!/bin/bash
# Experiments on the three-lane step: hardware queues x lanes, IIR wave priority, IIRs on one shared (CU-masked) stream.
mkdir -p gpurun_out/r03/exp
run() {   # name, env..., -- bench args
    local name=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" timeout -k 10 90 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/r03/exp/$name.json 2> gpurun_out/r03/exp/$name.err
    local rc=$?
    python - "$name" $rc <<'PY'
import json,sys
name,rc=sys.argv[1],sys.argv[2]
try:
    d=json.loads(open(f"gpurun_out/r03/exp/{name}.json").read().strip().splitlines()[-1])
    k=d["kernel_ms"]
    print(f"{name:34s} {d['value']/1e6:6.3f} M  steady {d.get('steady_state',{}).get('value',0)/1e6:6.3f} M  iir {k['iir']:.2f} pair {k['pair']:.2f} exact {k['exact']:.2f} spec {k['spec']:.2f}", flush=True)
except Exception as e:
    print(name, "rc", rc, "unreadable", e, flush=True)
PY
    return $rc
}
run base_q4_l3 X=1 -- || exit 1
run q8_l3 GPU_MAX_HW_QUEUES=8 -- || exit 1
run q8_l4 GPU_MAX_HW_QUEUES=8 -- --lanes 4 || exit 1
run q8_l5 GPU_MAX_HW_QUEUES=8 -- --lanes 5 || exit 1
run q8_l6 GPU_MAX_HW_QUEUES=8 -- --lanes 6 || exit 1
run q4_l4 X=1 -- --lanes 4 || exit 1
run prio1 ECHO_IIR_PRIO=1 -- || exit 1
run prio3 ECHO_IIR_PRIO=3 -- || exit 1
run side ECHO_IIR_SIDE=1 -- || exit 1
run side_cu192 ECHO_IIR_SIDE=1 ECHO_IIR_CUS=192 -- || exit 1
run side_cu128 ECHO_IIR_SIDE=1 ECHO_IIR_CUS=128 -- || exit 1
run side_cu224 ECHO_IIR_SIDE=1 ECHO_IIR_CUS=224 -- || exit 1
run side_prio3 ECHO_IIR_SIDE=1 ECHO_IIR_PRIO=3 -- || exit 1
run q8_l4_prio3 GPU_MAX_HW_QUEUES=8 ECHO_IIR_PRIO=3 -- --lanes 4 || exit 1
