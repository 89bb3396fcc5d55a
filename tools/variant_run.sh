#!/bin/bash
# bench.py in the timed mode and with one slice in flight for each named variant library (tools/build_variant.sh): tools/variant_run.sh base NAME ...
V=$PWD/deep-cine-cardiac-mri_amd/csrc/build/variants
CFG=${CFG:-2}
for v in "$@"; do
  if [ $v = base ]; then unset CINE_HIP_LIB; else export CINE_HIP_LIB=$V/libcine_hip_$v.so; fi
  for inf in 0 1; do
    timeout -k 10 200 python3 bench.py --config $CFG --steps 24 --warmup 3 --no-cpu-baseline --repeats 1 --headline-only --inflight $inf > gpurun_out/var_${v}_$inf.json 2> gpurun_out/var_${v}_$inf.err || { echo "$v $inf failed"; tail -3 gpurun_out/var_${v}_$inf.err; }
    python3 - <<P
import json
try:
    d=json.load(open("gpurun_out/var_${v}_$inf.json"))
    print("$v inflight=$inf value=%.1f ms=%.3f"%(d["value"],d["ms_per_step"]), {k:round(x,3) for k,x in d.get("kernel_ms_per_slice",{}).items() if x}, "err=%.2e"%d.get("parity_max_rel_err_vs_cpu_oracle",-1))
except Exception as e: print("$v $inf", e)
P
  done
done
