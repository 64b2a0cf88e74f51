#!/bin/bash
# Variants of the real-transform matched filter (thread count): bench of the airSonar.py set under three lanes and alone.
set -o pipefail
mkdir -p gpurun_out/r03/var
cp dsp_amd/libechohip.so /tmp/libechohip_base.so
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/libechohip_base.so dsp_amd/libechohip.so; else cp dsp_amd/variants/libecho_$v.so dsp_amd/libechohip.so; fi
  for lanes in 3 1; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --workload original --lanes $lanes > gpurun_out/r03/var/${v}_l$lanes.json 2> gpurun_out/r03/var/${v}_l$lanes.err || { echo "$v failed"; tail -5 gpurun_out/r03/var/${v}_l$lanes.err; cp /tmp/libechohip_base.so dsp_amd/libechohip.so; exit 1; }
    python - $v $lanes <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03/var/%s_l%s.json"%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[1], "lanes", sys.argv[2], "%.3f M"%(d["value"]/1e6), "ms/step %.3f"%d["ms_per_step"], {k: round(v,3) for k,v in d["kernel_ms"].items()})
PY
  done
done
cp /tmp/libechohip_base.so dsp_amd/libechohip.so
