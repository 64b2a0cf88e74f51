#!/bin/bash
# Duration of the fused full-chain kernel when it leaves after stage ECHO_ENV_STOP (20 FIR, 21 correlation transforms,
# 6 correlation stored, 1 / 2 Hilbert forward / inverse, 10 smoothing, 11 / 3 statistics, 12 / 13 candidates, 4 distance
# rule, 5 prominence, 0 everything): the band stage's time of an unoverlapped bench run.
for st in 20 21 6 1 2 10 11 3 12 13 4 5 0; do
  ECHO_ENV_STOP=$st timeout -k 10 100 python bench.py --no-cpu-baseline --mode full --lanes 3 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stop $st band stage %.3f ms' % d['roofline']['dominant_kernel_alone']['kernel_ms_all']['band'])" || exit 1
done
