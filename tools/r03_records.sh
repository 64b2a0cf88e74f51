#!/bin/bash
# Secondary bench records of round 3 (the variants of the headline command and the other parameter sets).
mkdir -p gpurun_out/r03
b() { local name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/$name.json 2> gpurun_out/r03/$name.err || { echo "$name rc=$?"; tail -3 gpurun_out/r03/$name.err; return 1; }
      python - "$name" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r03/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} {d['value']/1e6:7.3f} M  ms/step {d['ms_per_step']:.3f}  steps {d['steps']}", flush=True)
PY
}
b bench_1M_frames_sharded_job --total-frames 1000000 --steps 5 --warmup 1 || exit 1
b bench_1M_frames_sharded_job_force_dist --total-frames 1000000 --steps 5 --warmup 1 --force-dist || exit 1
b bench_no_spectra --no-spectra || exit 1
b bench_unoverlapped --no-pipeline || exit 1
b bench_full_mode --mode full || exit 1
b bench_full_mode_driver_protocol --mode full --steps 20 --warmup 5 || exit 1
ECHO_FCE=0 b bench_full_mode_two_kernels --mode full || exit 1
ECHO_ENV_LIN=0 b bench_full_mode_half_length_transforms --mode full || exit 1
b bench_original_variant --workload original || exit 1
b bench_original_variant_driver_protocol --workload original --steps 20 --warmup 5 || exit 1
b bench_twopeak_variant --workload twopeak --lanes 1 || exit 1
b bench_realtime_variant --workload realtime || exit 1
b bench_device_synth --device-synth || exit 1
