#!/bin/bash
# development aid: GPU tests of the chain, unpipelined and pipelined bench lines, kernel stats
set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_runloop.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > gpurun_out/tq.log 2>&1 || { tail -30 gpurun_out/tq.log; exit 1; }
tail -2 gpurun_out/tq.log
for flag in "--no-pipeline" ""; do
timeout -k 10 200 python bench.py --steps 30 --no-cpu-baseline $flag > gpurun_out/bq$flag.json 2>gpurun_out/bq.err
python -c "
import json; d=json.load(open('gpurun_out/bq$flag.json')); print('$flag', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms'].items()})"
done
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q -- python3 $R/bench.py --steps 20 --no-cpu-baseline --no-pipeline > /dev/null 2>&1
cd $R; f=$(ls -t $(find gpurun_out/prof_q -name "*kernel_stats.csv") | head -1); head -7 $f | cut -d, -f1,2,4
