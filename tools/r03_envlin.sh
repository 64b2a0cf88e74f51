#!/bin/bash
# Envelope kernel with the Hilbert transform as one 4608-point real convolution against the half-length transforms.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_lanes.py -x -q --timeout 180 --timeout-method thread -k "env or full or dropin or lanes" > gpurun_out/r03/envlin_tests.log 2>&1 || { tail -40 gpurun_out/r03/envlin_tests.log; exit 1; }
tail -1 gpurun_out/r03/envlin_tests.log
for lin in 0 1; do
  for st in 6 1 2 0; do
    ECHO_ENV_LIN=$lin ECHO_ENV_STOP=$st timeout -k 10 100 python tools/env_stage_times.py 2>/dev/null | sed "s/^/lin $lin /" || exit 1
  done
  ECHO_ENV_LIN=$lin timeout -k 10 200 python bench.py --no-cpu-baseline --mode full > gpurun_out/r03/full_lin$lin.json 2> gpurun_out/r03/full_lin$lin.err || { tail gpurun_out/r03/full_lin$lin.err; exit 1; }
  ECHO_ENV_LIN=$lin timeout -k 10 200 python bench.py --no-cpu-baseline --mode full --lanes 1 > gpurun_out/r03/full_l1_lin$lin.json 2> gpurun_out/r03/full_l1_lin$lin.err || exit 1
  python - $lin <<'PY'
import json,sys
for f in ("full_lin","full_l1_lin"):
    d=json.loads(open("gpurun_out/r03/%s%s.json"%(f,sys.argv[1])).read().strip().splitlines()[-1])
    print(f,sys.argv[1],"%.3f M"%(d["value"]/1e6),"ms/step %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["kernel_ms"].items()}, d["roofline"].get("dominant_kernel_alone",{}).get("kernel_ms_all"))
PY
done
