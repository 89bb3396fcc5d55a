#!/bin/bash
# tools/ab_variants.sh NAME ...: bench.py (headline only) for each variant library of csrc/build/variants ("base" = the product library),
# timed mode (inflight auto) and optionally one slice in flight (INF="0 1")
V=$PWD/deep-cine-cardiac-mri_amd/csrc/build/variants
CFG=${CFG:-2}; INF=${INF:-0}; STEPS=${STEPS:-36}
for v in "$@"; do
  if [ $v = base ]; then unset CINE_HIP_LIB; else export CINE_HIP_LIB=$V/libcine_hip_$v.so; fi
  for inf in $INF; do
    timeout -k 10 200 python3 bench.py --config $CFG --steps $STEPS --warmup 3 --no-cpu-baseline --repeats 2 --headline-only --inflight $inf > gpurun_out/var_${v}_$inf.json 2> gpurun_out/var_${v}_$inf.err || { echo "$v $inf failed"; tail -3 gpurun_out/var_${v}_$inf.err; }
    python3 - <<P
import json
try:
    d=json.load(open("gpurun_out/var_${v}_$inf.json"))
    print("$v inflight=$inf value=%.1f rep=%s"%(d["value"],[round(x,1) for x in d["repeat_values"]]), {k:round(x,3) for k,x in d.get("kernel_ms_per_slice",{}).items() if x})
except Exception as e: print("$v $inf", e)
P
  done
done
