#!/bin/bash
# Probe: the IIR in the lane mapping (half the instruction issue, 64 compute units, long latency) under 3 and 4 batches in
# flight, with the as-benched kernel durations -- do the other kernels un-stretch when the IIR leaves most of the chip?
mkdir -p gpurun_out/r03/exp
for l in 3 4; do
  for m in 0 1; do
    ECHO_IIR_MODE=$m timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --lanes $l > gpurun_out/r03/exp/lanemap_m${m}_l${l}.json 2> gpurun_out/r03/exp/lanemap_m${m}_l${l}.err || exit 1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/exp/lanemap_*.json')):
    d=json.load(open(f))
    print(f.split('/')[-1], round(d['value']/1e6,3), 'steady', round(d['steady_state']['value']/1e6,3) if isinstance(d.get('steady_state'),dict) else d.get('steady_state'), {k:round(v,3) for k,v in d['kernel_ms'].items()})
PY
