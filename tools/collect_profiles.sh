#!/bin/bash
# Collects the evidence behind bench.py's roofline object on a GPU box: rocprofv3 kernel-trace summaries of the bench
# command (as it runs by default, and with nothing overlapped), PMC passes for HBM bytes (FETCH_SIZE / WRITE_SIZE) and
# for the SQ counters behind the issue-busy and LDS figures, each pass in its own process.
# Usage (repo root): bash tools/collect_profiles.sh <outdir> [extra bench flags]
set -u
OUT=${1:-gpurun_out/prof}; shift || true
EXTRA="$*"
mkdir -p "$OUT"
R=$PWD
export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA"
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -- $BENCH > "$R/$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$R/$OUT/trace.log"; exit 1; }
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace_serial" -- $BENCH --no-pipeline > "$R/$OUT/trace_serial.log" 2>&1 || { echo "serial trace failed"; tail -5 "$R/$OUT/trace_serial.log"; exit 1; }
PMCBENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline $EXTRA"
pass() {  # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$R/$OUT/$1" -- $PMCBENCH > "$R/$OUT/$1.log" 2>&1 || { echo "pass $1 failed"; tail -5 "$R/$OUT/$1.log"; return 1; }
}
pass fetch "FETCH_SIZE" && pass write "WRITE_SIZE" &&
pass sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" &&
pass sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
cd "$R"
python3 tools/summarise_profiles.py "$OUT"
