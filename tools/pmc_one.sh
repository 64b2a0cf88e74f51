#!/bin/bash
# One rocprofv3 PMC pass over a short bench run. Usage: bash tools/pmc_one.sh <outdir> "<counters>"
OUT=$1; shift
mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d "$OUT" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
python tools/pmc_summary.py "$OUT"
