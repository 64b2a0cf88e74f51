#!/bin/bash
# Collects the evidence behind bench.py's roofline object on a GPU box: a rocprofv3 kernel-trace summary of the
# bench command and two PMC passes (HBM fetch / write bytes), each in its own process.
# Usage (repo root): bash tools/collect_profiles.sh <outdir>
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.log" 2>&1 || { echo "fetch failed"; tail -5 "$OUT/fetch.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.log" 2>&1 || { echo "write failed"; tail -5 "$OUT/write.log"; exit 1; }
python tools/summarise_profiles.py "$OUT"
