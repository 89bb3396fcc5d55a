#!/bin/bash
# tools/ab_mask.sh MASK ...: bench.py headline-only per cine_set_conv_plane mask (bit 0 plane-wide 3x3, bit 1 transpose convs, bit 2 wide planes / volumes: 7 = all lean kernels, 0 = general kernels)
CFG=${CFG:-2}; INF=${INF:-0}; STEPS=${STEPS:-40}
for m in "$@"; do
  for inf in $INF; do
    timeout -k 10 200 python3 bench.py --config $CFG --steps $STEPS --warmup 3 --no-cpu-baseline --repeats 2 --headline-only --inflight $inf --conv-plane-mask $m > gpurun_out/mask_${m}_$inf.json 2> gpurun_out/mask_${m}_$inf.err || { echo "$m $inf failed"; tail -3 gpurun_out/mask_${m}_$inf.err; }
    python3 - <<P
import json
try:
    d=json.load(open("gpurun_out/mask_${m}_$inf.json"))
    print("mask=$m inflight=$inf value=%.1f rep=%s"%(d["value"],[round(x,1) for x in d["repeat_values"]]), {k:round(x,3) for k,x in d.get("kernel_ms_per_slice",{}).items() if x})
except Exception as e: print("$m $inf", e)
P
  done
done
