#!/bin/bash
# tools/ab_cfgs.sh "MASKS" "CFGS": bench.py headline-only for every (config, cine_set_conv_plane mask) pair
for c in $2; do for m in $1; do
  timeout -k 10 250 python3 bench.py --config $c --steps 24 --warmup 3 --no-cpu-baseline --repeats 1 --headline-only --conv-plane-mask $m > gpurun_out/cfg${c}_m$m.json 2> gpurun_out/cfg${c}_m$m.err || { echo "cfg $c mask $m failed"; tail -3 gpurun_out/cfg${c}_m$m.err; }
  python3 - <<P
import json
try:
    d=json.load(open("gpurun_out/cfg${c}_m$m.json"))
    print("cfg=$c mask=$m value=%.1f rep=%s"%(d["value"],[round(x,1) for x in d["repeat_values"]]), {k:round(x,3) for k,x in d.get("kernel_ms_per_slice",{}).items() if x})
except Exception as e: print("cfg $c mask $m", e)
P
done; done
