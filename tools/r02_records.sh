#!/bin/bash
# Produces the secondary measurement records kept under profiles/r02/ in one GPU call. Usage: bash tools/r02_records.sh <outdir>
set -u
OUT=${1:-gpurun_out/r02}; mkdir -p "$OUT"; R=$PWD; export TMPDIR=/tmp
b() { name=$1; shift; timeout -k 10 240 python3 bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "$name failed"; tail -3 "$OUT/$name.err"; return 1; }
      python3 -c "import json; d=json.load(open('$OUT/$name.json')); print('$name', round(d['value']), round(d['ms_per_step'],3))"; }
b bench_default &&
b bench_unoverlapped --no-pipeline --no-cpu-baseline &&
b bench_one_stream_prefetch --lanes 1 --no-cpu-baseline &&
b bench_force_dist_one_rank --force-dist --no-cpu-baseline &&
b bench_1M_frames_sharded_job --total-frames 1000000 --steps 5 --warmup 1 &&
b bench_full_mode --mode full --lanes 1 --no-cpu-baseline &&
b bench_original_variant --workload original --lanes 3 --no-cpu-baseline &&
b bench_twopeak_variant --workload twopeak --lanes 1 --no-cpu-baseline &&
b bench_realtime_variant --workload realtime --lanes 3 --no-cpu-baseline &&
b bench_distinct_bands_no_spectra --no-spectra --no-cpu-baseline &&
timeout -k 10 200 python3 tools/stream_latency.py > "$OUT/stream_latency_realtime.json" 2> "$OUT/stream_latency.err" &&
cd /tmp &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace_stream" -- python3 $R/tools/stream_latency.py > /dev/null 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace_twopeak" -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload twopeak --no-pipeline > /dev/null 2>&1
cd $R
for t in stream twopeak; do f=$(ls -t $(find $OUT/trace_$t -name "*kernel_stats.csv") | head -1); python3 - "$f" "$OUT/kernel_stats_$t.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "echo" in r["Name"]]
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader()
    for r in rows:
        r["Name"] = r["Name"].split("(")[0]; w.writerow(r)
        print(" ", r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
