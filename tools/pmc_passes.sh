#!/bin/bash
# Collects rocprofv3 PMC counters for one short bench run, one counter group per pass (each in its own process).
# Usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$OUT/$1" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/$1.log" 2>&1 || { echo "pass $1 failed"; tail -5 "$OUT/$1.log"; return 1; }
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" &&
run sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" &&
run fetch "FETCH_SIZE" &&
run write "WRITE_SIZE" &&
run grbm "GRBM_GUI_ACTIVE"
