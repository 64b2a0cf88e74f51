#!/bin/bash
# Kernel durations of the long-frame set, one batch at a time (no overlap), with the real transform and without.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03/rcfprof
for mode in 0 3; do
  export ECHO_RCF=$mode
  if [ $mode = 2 ]; then export ECHO_RCF=1 ECHO_FUSE_DET=0; fi
  if [ $mode = 3 ]; then export ECHO_RCF=2; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03/rcfprof/m$mode -o out -- python3 $R/bench.py --workload ${1:-original} --lanes 1 --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r03/rcfprof/m$mode.log 2>&1 || { tail -5 $R/gpurun_out/r03/rcfprof/m$mode.log; exit 1; }
  echo "mode $mode (0 two sub-blocks in one 8192-point transform; 1 one real transform; 2 the same, detector as a kernel of its own; 3 real transforms per block of lags, two workgroups per CU, detector as a kernel of its own)"
  python3 - $R/gpurun_out/r03/rcfprof/m$mode/out_results.db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "info_kernel" in t or "kernel_symbol" in t][0]
for r in db.execute("select s.kernel_name, count(*), avg(d.end-d.start)/1e6 from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc limit 6" % (kd, ks)):
    print("  %-72s %4d %.3f ms" % (r[0][:72], r[1], r[2]))
PY
done
unset ECHO_FUSE_DET
cd $R
for mode in 0 2; do
  ECHO_RCF=$mode timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload ${1:-original} > gpurun_out/r03/rcfprof/chain_m$mode.json 2> gpurun_out/r03/rcfprof/chain_m$mode.err || exit 1
  python3 - $mode <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03/rcfprof/chain_m%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("chain ECHO_RCF=%s"%sys.argv[1], "%.3f M"%(d["value"]/1e6), "ms/step %.3f"%d["ms_per_step"])
PY
done
