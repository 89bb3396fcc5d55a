#!/bin/bash
# A/B of the lean plane kernel on the GPU box: bench.py headline-only with the lean kernel on / off, timed mode and one slice in flight
CFG=${CFG:-2}
for v in lean general; do
  flag=""; [ $v = general ] && flag="--no-conv-plane"
  for inf in 0 1; do
    timeout -k 10 200 python3 bench.py --config $CFG --steps 24 --warmup 3 --no-cpu-baseline --repeats 1 --headline-only --inflight $inf $flag > gpurun_out/ab_${v}_$inf.json 2> gpurun_out/ab_${v}_$inf.err || { echo "$v $inf failed"; tail -3 gpurun_out/ab_${v}_$inf.err; }
    python3 - <<P
import json
try:
    d=json.load(open("gpurun_out/ab_${v}_$inf.json"))
    print("$v inflight=$inf value=%.1f ms=%.3f"%(d["value"],d["ms_per_step"]), {k:round(x,3) for k,x in d.get("kernel_ms_per_slice",{}).items() if x})
except Exception as e: print("$v $inf", e)
P
  done
done
