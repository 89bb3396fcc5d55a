for q in 16 32; do for inf in 6 10 15; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --repeats 2 --inflight $inf > gpurun_out/inf_${q}_$inf.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('gpurun_out/inf_${q}_$inf.json'))
print('queues=$q inflight=$inf value=%.1f'%d['value'], [round(v,1) for v in d['repeat_values']])
"
done; done
