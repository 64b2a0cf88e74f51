#!/bin/bash
set -o pipefail
for st in 6 1 2 10 11 3 12 13 4 5 0; do
  ECHO_ENV_STOP=$st timeout -k 10 100 python tools/env_stage_times.py 2>/dev/null || exit 1
done
